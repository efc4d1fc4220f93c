// INR forward on the CDNA4 matrix cores — replaces the per-voxel MLP evaluation of the reference
// (inr/inr/model.py:11-50 build_input + apply_mlp, :119-141 predict_volume; SIREN forward of
// notebooks/neumors_inr.ipynb:1165-1178), which the viewer runs as a prepass over every voxel
// (inr/viewer/brats_viewer.py:250-310).
//
// Formulation: every layer is computed TRANSPOSED,  Y^T[out, points] = W^T[out, in] . H^T[in, points],
// with v_mfma_f32_32x32x16_bf16.  The 32x32 accumulator tile then has the point on the lane and the
// output feature in the registers — exactly the B-operand shape of the next layer (k = feature), so
// activations go  accumulator -> bias + activation -> bf16 pack -> next MFMA  without leaving the
// register file (guide: "an accumulator tile as the next MFMA's operand").  The k order inside a
// 16-deep step is permuted by that reuse (element j of lane half h is feature 16s + 8(j>>2) + 4h + (j&3));
// the weights are pre-packed in the same permuted order (mrirt_inr_pack_weights), one 1-KiB fragment
// per (out tile, k step): an A fragment is one 16-byte read per lane.
//
// Work split: a workgroup is 8 waves (2 per SIMD: one wave's DMA / LDS waits are covered by the other's
// MFMAs); each wave owns 32 points of a 256-point batch, and the workgroup is PERSISTENT: it walks batches
// round-robin and the weight stream is cyclic (the head chunk stages the next batch's first chunk), so
// launch, bias staging and DMA latency are paid once.  The weights of up to four out tiles (<= 64 KiB) are a
// "chunk": chunks are consecutive in the packed image and are staged one chunk ahead by LDS-DMA
// (global_load_lds, one 1-KiB fragment per wave-instruction into a lane-linear image; double buffer,
// one barrier per chunk; the pieces are dealt through the MFMA stream, right after each tile's bias read),
// then read by all 8 waves with conflict-free ds_read_b128 through a four-deep register ring.  The
// accumulator starts at the bias and the layer scale is folded into the packed weights, so an activation is
// v_sin_f32 (or half a v_pk_max_f32) plus the packed bf16 convert, sliced under the next tile's MFMAs.
// Layer-0 inputs: a per-workgroup LDS table of feature descriptors applied to the wave's staged raw inputs.
//
// Precision: bf16 operands, fp32 accumulate.  A SIREN's FIRST layer scales its inputs by w0 = 30 before a
// sine, so its inputs and weights are split hi + lo in bf16 and three products are accumulated
// (hi*hi + hi*lo + lo*hi: ~16 mantissa bits).  A ReLU net's first layer (coordinates, sin(pi k c) features,
// z-scored intensities) and every hidden layer use plain bf16.
#include <atomic>
#include <stdlib.h>

#include "mrirt_host.h"

namespace mrirt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N> struct IC { static constexpr int value = N; };   // compile-time int as a lambda argument

constexpr int kInrWaves = 8;      // waves per workgroup
constexpr int kMaxLayers = 8;
constexpr int kPackSlackFrags = 64;   // one full chunk
constexpr int kRefSlackFrags = 32;    // after the refine image: inr_refine_kernel's tile DMAs always move a full ring slot
constexpr int kRawStride = 12;        // floats per point in the LDS copy of the raw inputs: c0..c2, m0..m7, 0
constexpr int kMaxMods = 8;

struct InrLayout {
    uint32_t numLayers, hidden, inDim, outDim, kt0;      // kt0: 32-wide k tiles of layer 0
    uint32_t in[kMaxLayers], out[kMaxLayers];            // true widths
    uint32_t fragOff[kMaxLayers];                        // first fragment (1 KiB units) of each layer's image
    uint32_t biasOff[kMaxLayers];                        // offset into the padded bias array
    uint32_t wOff[kMaxLayers];                           // offset into the unpadded fp32 weight array
    uint32_t totalFrags;
    uint32_t split0;                                     // layer 0 carries hi + lo fragments (SIREN kinds)
    uint32_t aug0;                                       // ... or, for <= 8 inputs, the split lives in the k axis (below)
    // Folded into the packed weights and the LDS copy of the biases, so that the activation is one
    // instruction on the accumulator: SIREN layers are sin(2 pi . rev) with v_sin_f32 taking revolutions,
    // hence scale = w0 / 2 pi (layer 0), 1 / 2 pi (hidden), 1 (head); ReLU nets: 1 everywhere.
    float scale[kMaxLayers];
    float bscale[kMaxLayers];                            // the biases' share of it: 1 / 2 pi without the w0
    // The REFINE image (near-tie refinement, inr_refine_kernel): every layer as split bf16, fragments
    // [out tile][k step][hi, lo] (the augmented layer 0 keeps its k-folded form [out tile][k step]); it follows
    // the image above and its slack.  refFragOff is relative to the refine image's first fragment.
    uint32_t refFragOff[kMaxLayers];
    uint32_t refTotalFrags;
};
// one more fragment after the refine image: the calibration record (tail[0] = rms logit error of the bf16 pass)

// Packed image, in fragment units:  layer 0 of a SIREN: [o][t][s][hi,lo]   every other layer: [o][t][s]
// Augmented split (a SIREN with <= 8 inputs, i.e. the 7-input net of the notebook): the three products
// x_hi W_hi + x_lo W_hi + x_hi W_lo are folded into the k axis of ONE 32-deep tile — k step 0 carries
// [x_hi | x_lo] against [W_hi ; W_hi], k step 1 carries [x_hi] against [W_lo] — two MFMAs per out tile
// instead of six.
static int make_layout(const MrirtInrDesc* d, InrLayout& L) {
    if (!d) return MRIRT_ERR_NULL;
    if (d->numLayers < 2 || d->numLayers > kMaxLayers || d->outDim < 1 || d->outDim > 16) return MRIRT_ERR_ARG;
    if (d->hidden != 32 && d->hidden != 64 && d->hidden != 128 && d->hidden != 256) return MRIRT_ERR_ARG;
    if (d->inDim < 1 || d->inDim > 128 || d->kind > 3) return MRIRT_ERR_ARG;
    if (d->kind == MRIRT_INR_FOURIER_RELU && d->inDim != 3 + 6 * d->fourierFreqs + d->numMods) return MRIRT_ERR_ARG;
    if (d->kind == MRIRT_INR_SIREN && d->inDim != 3 + d->numMods) return MRIRT_ERR_ARG;
    if (d->kind < 2 && d->numMods > (uint32_t)kMaxMods) return MRIRT_ERR_ARG;    // raw inputs are staged 12 floats per point
    L.numLayers = d->numLayers; L.hidden = d->hidden; L.inDim = d->inDim; L.outDim = d->outDim;
    L.kt0 = d->inDim <= 32 ? 1 : 4;                      // 33..128 inputs: zero-padded to 128 (layer 0 only)
    uint32_t frag = 0, bias = 0, w = 0;
    for (uint32_t l = 0; l < d->numLayers; ++l) {
        L.in[l] = l == 0 ? d->inDim : d->hidden;
        L.out[l] = l + 1 == d->numLayers ? d->outDim : d->hidden;
        const uint32_t kt = l == 0 ? L.kt0 : d->hidden / 32, ot = (L.out[l] + 31) / 32;
        L.fragOff[l] = frag; L.biasOff[l] = bias; L.wOff[l] = w;
        const bool sirenNet = d->kind == MRIRT_INR_SIREN || d->kind == 3u, head = l + 1 == d->numLayers;
        L.scale[l] = (!sirenNet || head) ? 1.0f : (float)((l == 0 ? (double)d->w0 : 1.0) / 6.283185307179586);
        L.bscale[l] = (!sirenNet || head) ? 1.0f : (float)(1.0 / 6.283185307179586);     // sin(w0 (xW) + b): b is not scaled by w0
        const bool aug = d->kind == MRIRT_INR_SIREN && d->inDim <= 8;
        frag += ot * kt * 2 * ((l == 0 && sirenNet && !aug) ? 2 : 1);     // split layer 0: hi and lo fragments side by side
        bias += ot * 32;
        w += L.in[l] * L.out[l];
    }
    L.totalFrags = frag;
    L.aug0 = (d->kind == MRIRT_INR_SIREN && d->inDim <= 8) ? 1u : 0u;
    L.split0 = ((d->kind == MRIRT_INR_SIREN || d->kind == 3u) && !L.aug0) ? 1u : 0u;
    uint32_t rf = 0;
    for (uint32_t l = 0; l < d->numLayers; ++l) {
        const uint32_t kt = l == 0 ? L.kt0 : d->hidden / 32, ot = (L.out[l] + 31) / 32;
        L.refFragOff[l] = rf;
        rf += ot * kt * 2 * ((l == 0 && L.aug0) ? 1 : 2);
    }
    L.refTotalFrags = rf;
    return MRIRT_OK;
}
// first fragment of the refine image / of the calibration record inside the packed buffer
static inline uint32_t ref_base_frag(const InrLayout& L) { return L.totalFrags + (uint32_t)kPackSlackFrags; }
__device__ __forceinline__ uint32_t ref_base_frag_dev(const InrLayout& L) { return L.totalFrags + (uint32_t)kPackSlackFrags; }
static inline uint32_t tail_frag(const InrLayout& L) { return ref_base_frag(L) + L.refTotalFrags + (uint32_t)kRefSlackFrags; }

__device__ __forceinline__ uint16_t bf16_bits(float x) {     // round-to-nearest-even, NaN preserved by the cast
    return __builtin_bit_cast(uint16_t, (__bf16)x);
}

// fp32 [in][out] row-major -> permuted bf16 fragments (see file header).  One thread per (fragment, lane).
__global__ __launch_bounds__(256) void inr_pack_kernel(const float* __restrict__ w, uint4* __restrict__ packed, InrLayout L) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t frag = gid >> 6, lane = gid & 63u;
    if (frag >= L.totalFrags) return;
    uint32_t l = 0;
    while (l + 1 < L.numLayers && frag >= L.fragOff[l + 1]) ++l;
    const uint32_t kt = l == 0 ? L.kt0 : L.hidden / 32;
    uint32_t f = frag - L.fragOff[l];
    bool lo = false;
    if (l == 0 && L.split0) { lo = (f & 1u) != 0; f >>= 1; }
    const uint32_t s = f & 1u, t = (f >> 1) % kt, o = (f >> 1) / kt;
    const uint32_t r = lane & 31u, h = lane >> 5;
    uint16_t e[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        uint32_t k = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3u);
        const uint32_t oc = 32 * o + r;
        bool part_lo = lo;
        if (l == 0 && L.aug0) {                          // k step 0: [W_hi ; W_hi], k step 1: [W_lo]
            const uint32_t kk = k & 15u, in = L.in[0];
            part_lo = s == 1;
            k = s == 0 ? (kk < in ? kk : (kk < 2 * in ? kk - in : 0xffffu)) : (kk < in ? kk : 0xffffu);
        }
        const float v = (k < L.in[l] && oc < L.out[l]) ? w[L.wOff[l] + k * L.out[l] + oc] * L.scale[l] : 0.0f;
        const uint16_t hi = bf16_bits(v);
        const float hif = __builtin_bit_cast(float, (uint32_t)hi << 16);
        e[j] = part_lo ? bf16_bits(v - hif) : hi;
    }
    uint4 q;
    q.x = e[0] | ((uint32_t)e[1] << 16); q.y = e[2] | ((uint32_t)e[3] << 16);
    q.z = e[4] | ((uint32_t)e[5] << 16); q.w = e[6] | ((uint32_t)e[7] << 16);
    packed[(size_t)frag * 64 + lane] = q;
}

// The refine image: one thread per (fragment, lane).  Same permuted k order and the same folded scale as above;
// fragment 2q holds the bf16 "hi" parts and 2q + 1 the "lo" parts (w - hi, rounded to bf16) of k step q.
__global__ __launch_bounds__(256) void inr_pack_ref_kernel(const float* __restrict__ w, uint4* __restrict__ ref, InrLayout L) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t frag = gid >> 6, lane = gid & 63u;
    if (frag >= L.refTotalFrags) return;
    uint32_t l = 0;
    while (l + 1 < L.numLayers && frag >= L.refFragOff[l + 1]) ++l;
    const uint32_t kt = l == 0 ? L.kt0 : L.hidden / 32;
    uint32_t f = frag - L.refFragOff[l];
    const bool aug = l == 0 && L.aug0;
    bool lo = false;
    if (!aug) { lo = (f & 1u) != 0; f >>= 1; }
    const uint32_t s = f & 1u, t = (f >> 1) % kt, o = (f >> 1) / kt;
    const uint32_t r = lane & 31u, h = lane >> 5;
    uint16_t e[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        uint32_t k = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3u);
        const uint32_t oc = 32 * o + r;
        bool part_lo = lo;
        if (aug) {                                       // k step 0: [W_hi ; W_hi], k step 1: [W_lo]  (as inr_pack_kernel)
            const uint32_t kk = k & 15u, in = L.in[0];
            part_lo = s == 1;
            k = s == 0 ? (kk < in ? kk : (kk < 2 * in ? kk - in : 0xffffu)) : (kk < in ? kk : 0xffffu);
        }
        const float v = (k < L.in[l] && oc < L.out[l]) ? w[L.wOff[l] + k * L.out[l] + oc] * L.scale[l] : 0.0f;
        const uint16_t hi = bf16_bits(v);
        const float hif = __builtin_bit_cast(float, (uint32_t)hi << 16);
        e[j] = part_lo ? bf16_bits(v - hif) : hi;
    }
    uint4 q;
    q.x = e[0] | ((uint32_t)e[1] << 16); q.y = e[2] | ((uint32_t)e[3] << 16);
    q.z = e[4] | ((uint32_t)e[5] << 16); q.w = e[6] | ((uint32_t)e[7] << 16);
    ref[(size_t)frag * 64 + lane] = q;
}

// Near-tie refinement.  bf16 hidden layers leave each logit with an error of a few 1e-3 of the logit
// range; where the two largest logits are closer than that, the argmax can differ from the fp32 reference's
// (0.3-0.5 % of the points of a random 4-class head).  The forward kernels therefore mark every point whose top-2 gap is
// below  tieScale x (rms logit error of this network's bf16 pass, measured once at pack time: tail[0])  by setting
// kFlagBit in the class they store, and inr_refine_kernel re-evaluates exactly those points with split-bf16 operands
// (hi + lo, three products, ~16 mantissa bits) in EVERY layer and stores the clean class.  The mark depends on the
// point's own logits only, so results stay independent of batch position.
constexpr int kFlagBit = 0x4000;          // classes are < 16: bit 14 of the int16 is free
constexpr float kTieSigmas = 3.0f;        // mark below this many standard deviations of the gap's error (MRIRT_INR_TIE_SIGMAS overrides)
constexpr int kCalPoints = 8192;          // calibration points (pack time)

struct InrArgs {
    InrLayout L;
    uint32_t kind, K, M;
    float w0;
    const uint4* wpack;
    const float* bias;
    const float* coords;       // [n][3] or nullptr (volume mode / raw-x kinds)
    const float* feats;        // [n][M]  (points mode)  or mods[M][H*W*D] (volume mode)
    int64_t n;
    const uint32_t* nDev;      // optional: the point count lives on the device (<= n); lets a producer kernel size the batch
    uint32_t volume;           // 1: points are the voxels of an H x W x D grid in ij order
    uint32_t H, W, D;
    float* logits;
    int16_t* argmax;
    const float* tie;          // calibration record of the packed net (tail[0] = rms logit error); nullptr: no marking
    float tieScale;            // mark when (best - second) < tieScale * tie[0]
    uint32_t refineAll;        // inr_refine_kernel: every point, not only the marked ones (calibration, mrirt_inr_forward_refined)
    uint32_t* segTicket;       // inr_refine_kernel: optional zeroed device word — the 2048-point segments after each workgroup's first are
                               // dealt on demand instead of round-robin (the config-5 passes: a few batches per workgroup, where the
                               // spread of the mark counts costs a whole batch time)
    uint32_t flags;            // MrirtInrFlags of the descriptor (read by the launchers only)
    float tieSigmas;           // the descriptor's; 0 = kTieSigmas
};

// top-2 tracking for the near-tie mark: v joins (best, second); strict > keeps the first maximum (np.argmax)
__device__ __forceinline__ void top2_push(float v, uint32_t cls, float& best, float& second, uint32_t& bestc) {
    const bool take = v > best;
    second = take ? best : fmaxf(second, v);
    best = take ? v : best;
    bestc = take ? cls : bestc;
}

constexpr int kResidentFrags = 56;    // RES: the whole packed image (<= 56 KiB) lives in LDS; two workgroups still fit a CU

// RES (small nets: the 4 x 64 network of inr/interactive.ipynb is 36 fragments): the weights are loaded into LDS
// once per workgroup and stay; the batch loop then has no DMA and no barrier at all — with chunks this small the
// per-chunk barrier and the DMA latency behind it were most of a batch's time.
template <int HID, int KT0, bool SIREN, bool AUG = false, bool RES = false>
__global__ __launch_bounds__(512) void inr_forward_kernel(const InrArgs a) {
    constexpr int KT = HID / 32;                         // k tiles of a hidden-wide input == out tiles of a hidden-wide output
    constexpr int OTC = KT < 4 ? KT : 4;                 // out tiles per chunk (one barrier per chunk)
    // The first layer of a SIREN multiplies its inputs by w0 = 30 before a sine, so it runs in split bf16 (hi + lo
    // operands, three products: ~16 mantissa bits).  A ReLU net's first layer sees coordinates, sin / cos
    // features and z-scored intensities and is as tolerant of bf16 as its hidden layers: one product.
    // AUG: the split folded into the k axis (<= 8 inputs; see the packed-image note above).
    constexpr bool SPLIT = SIREN && !AUG;
    constexpr int F0 = KT0 * (SPLIT ? 4 : 2), FH = KT * 2;   // fragments per out tile: layer 0 / other layers
    constexpr int CH0 = OTC * F0, CHH = OTC * FH;        // fragments per chunk
    constexpr int CHMAX = CH0 > CHH ? CH0 : CHH;
    constexpr int PERW = (CHMAX + kInrWaves - 1) / kInrWaves;
    constexpr int RD = 4;                                // A-fragment prefetch distance (ring of registers)
    // DMA pieces of the NEXT chunk are dealt over the first three quarters of this chunk's k steps
    constexpr int SP0 = (OTC * KT0 * 2 * 3 / 4) / PERW > 0 ? (OTC * KT0 * 2 * 3 / 4) / PERW : 1;
    constexpr int SPH = (CHH * 3 / 4) / PERW > 0 ? (CHH * 3 / 4) / PERW : 1;
    constexpr int PPT = (PERW + OTC - 1) / OTC;          // hidden layers: pieces per out tile
    static_assert((PERW - 1) * SP0 < OTC * KT0 * 2 && (PERW - 1) * SPH < CHH, "every DMA piece must get a slot");
    constexpr int kBiasQ = kMaxLayers * 256 / 4;         // every layer's padded biases, as float4
    constexpr int kTabQ = 128;                           // one descriptor per layer-0 input feature
    constexpr int kRawQ = kInrWaves * 32 * kRawStride / 4;   // each wave's 32 points x (3 coords, <= 8 mods, a zero)
    // 2 weight buffers (<= 64 KiB each) + biases + feature table + raw inputs: ONE LDS object (G17)
    constexpr int kWQ = (RES ? kResidentFrags : 2 * CHMAX) * 64;     // weight region, in uint4
    __shared__ uint4 ldsAll[kWQ + kBiasQ + kTabQ + kRawQ];
    uint4 (*lds)[CHMAX * 64] = reinterpret_cast<uint4 (*)[CHMAX * 64]>(ldsAll);
    const float4* ldsBias = reinterpret_cast<const float4*>(ldsAll + kWQ);
    float4* ldsTab = reinterpret_cast<float4*>(ldsAll + kWQ + kBiasQ);
    float* ldsRaw = reinterpret_cast<float*>(ldsAll + kWQ + kBiasQ + kTabQ);
    // point count: a launch argument, or (C5's chunked render) a device word written by the producer kernel
    int64_t nPts = a.n;
    if (a.nDev != nullptr) { const int64_t nd = (int64_t)*a.nDev; nPts = nd < nPts ? nd : nPts; }
    if (nPts <= 0) return;                               // uniform: nothing was staged, no barrier is pending
    const float tieThr = a.tie != nullptr ? a.tie[0] * a.tieScale : 0.0f;
    // Feature table (inr/inr/model.py:11-23 order: coords, per axis [sin k=1..K, cos k=1..K], modalities):
    // feature f of a point is  trig == 1 ? sin(2 pi (raw[src] * mult + phase)) : raw[src]  (trig == 2: the bf16
    // remainder raw[src] - bf16(raw[src]) of the augmented split)  with raw = (c0,c1,c2,
    // m0..m7, 0).  sin(pi k c) / cos(pi k c) go through v_sin_f32, which takes revolutions: mult = k/2
    // (|.| <= 8 at K = 16), phase 0 / 0.25; its ~1e-6 absolute error is far below the split-bf16
    // resolution of the layer-0 operands.
    if (a.kind < 2 && threadIdx.x < 128) {
        const uint32_t f = threadIdx.x;
        uint32_t src = kRawStride - 1, trig = 0;
        float mult = 0.0f, phase = 0.0f;
        if (f < 3) src = f;
        else if (f < a.L.inDim) {
            uint32_t g = f - 3;
            if (a.kind == MRIRT_INR_FOURIER_RELU && g < 6 * a.K) {
                const uint32_t axis = g / (2 * a.K), rem = g % (2 * a.K);
                const bool isSin = rem < a.K;
                src = axis; trig = 1;
                mult = (float)((isSin ? rem : rem - a.K) + 1) * 0.5f;
                phase = isSin ? 0.0f : 0.25f;
            } else {
                if (a.kind == MRIRT_INR_FOURIER_RELU) g -= 6 * a.K;
                src = 3 + g;
            }
        }
        if constexpr (AUG) {                             // slot f = 16 s + kk:  s = 0: [x_hi | x_lo],  s = 1: [x_hi]
            const uint32_t kk = f & 15u, in = a.L.inDim;
            src = kRawStride - 1; trig = 0;
            if (f < 16) { if (kk < in) src = kk; else if (kk < 2 * in) { src = kk - in; trig = 2; } }
            else if (f < 32 && kk < in) src = kk;
        }
        ldsTab[f] = make_float4(mult, phase, __builtin_bit_cast(float, src), __builtin_bit_cast(float, trig));
    }
    {   // biases: global -> LDS once, so the loop's only vector-memory traffic is the weight LDS-DMA
        const uint32_t nq = (a.L.biasOff[a.L.numLayers - 1] + 32) / 4;
        for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) {
            uint32_t l = 0;
            while (l + 1 < a.L.numLayers && 4 * i >= a.L.biasOff[l + 1]) ++l;
            float4 b = reinterpret_cast<const float4*>(a.bias)[i];
            const float sc = a.L.bscale[l];                     // the fold of the packed weights, minus layer 0's w0
            b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
            ldsAll[kWQ + i] = __builtin_bit_cast(uint4, b);
        }
    }

    // lane / r / h are re-derived (opaquely) at the top of every batch: addresses computed from a
    // loop-invariant lane id would be hoisted out of the batch loop and spilled around it
    uint32_t lane = threadIdx.x & 63u, r = lane & 31u, h = lane >> 5;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ wp = a.wpack;
    const uint32_t waveS = __builtin_amdgcn_readfirstlane(wave);
    constexpr bool siren = SIREN;                        // sin activations (kinds 1, 3) vs ReLU (kinds 0, 2)

    // ---- chunk streaming: LDS-DMA, one 1-KiB fragment per wave-instruction (lane-linear image) -----------
    // piece i of this wave = fragment wave + 8 i of the chunk.  A DMA instruction costs its wave tens of
    // issue cycles, so inside the MFMA streams the pieces are dealt out one every few k steps (the other
    // wave of the SIMD keeps the matrix core busy meanwhile) instead of eight in a row at the chunk's start.
    // The fragment count is a compile-time constant (a chunk that is followed by the short head chunk
    // fetches a full hidden chunk: the packed image carries that much slack, mrirt_inr_pack_bytes), so
    // with 8 | N the piece is unconditional and the MFMA stream stays one basic block.
    auto stage_piece = [&](uint32_t fragStart, auto nfragC, int dstBuf, int i) {
        if constexpr (RES) return;
#ifdef MRIRT_EXP_NODMA
        return;                                          // timing experiment: no weight staging (results are garbage)
#endif
        constexpr int N = decltype(nfragC)::value;
        const int f = (int)waveS + i * kInrWaves;                    // scalar: the source is s[base] + lane * 16
        if ((N % kInrWaves == 0) ? (i < N / kInrWaves) : (f < N))
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(wp) + ((size_t)(fragStart + f) << 10) + (lane << 4)),
                (__attribute__((address_space(3))) void*)(&lds[dstBuf][f * 64]), 16, 0, 0);
    };
    auto stage_issue = [&](uint32_t fragStart, auto nfragC, int dstBuf) {
#pragma unroll
        for (int i = 0; i < PERW; ++i) stage_piece(fragStart, nfragC, dstBuf, i);
    };
    uint32_t curFrag = a.L.fragOff[0];                   // RES: first fragment of the chunk being read
    auto frag_at = [&](int buf, int f) {
#ifdef MRIRT_EXP_NOLDSREAD
        return __builtin_bit_cast(bf16x8, make_uint4(0x3c003c00u + f, 0x3c003c00u, lane, 0x3c003c00u));
#endif
        if constexpr (RES) return __builtin_bit_cast(bf16x8, ldsAll[(curFrag + f) * 64 + lane]);
        else return __builtin_bit_cast(bf16x8, lds[buf][f * 64 + lane]);
    };

    uint32_t nextFrag = a.L.fragOff[0];
    int buf = 0;
    if constexpr (RES) {
        for (uint32_t f = waveS; f < a.L.totalFrags; f += kInrWaves)     // the whole image, once
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(wp) + ((size_t)f << 10) + (lane << 4)),
                (__attribute__((address_space(3))) void*)(&ldsAll[f * 64]), 16, 0, 0);
    } else {
        stage_issue(nextFrag, IC<CH0>{}, 0);
    }
    nextFrag += CH0;

    // layer-0 B operands: this lane's point, features 16s + 8(j>>2) + 4h + (j&3) of k tile t
    bf16x8 xin_hi[KT0][2], xin_lo[KT0][2];
    // Every load below is unconditional and straight-line (clamped index, value masked afterwards): a load
    // inside a per-feature branch makes hipcc wait vmcnt(0) per feature — one memory round trip each.
    auto load_inputs = [&](int64_t pidx) {
#ifdef MRIRT_EXP_NOINPUT
        for (int t = 0; t < KT0; ++t) for (int s2 = 0; s2 < 2; ++s2) for (int j = 0; j < 8; ++j) { xin_hi[t][s2][j] = (__bf16)(float)(pidx & 7); xin_lo[t][s2][j] = (__bf16)0.0f; }
        return;
#endif
        const int64_t p = pidx < nPts ? pidx : nPts - 1; // clamp: compute something, store nothing
        if (a.kind < 2) {
            // lane half 0 stages its point's raw inputs (3 coords + <= 8 modalities + a zero) in LDS ...
            float* raw = ldsRaw + waveS * 32 * kRawStride + r;       // [input][point]: lanes of a half hit 32 banks
            if (h == 0) {
                float c[3], m[kMaxMods];
                const uint32_t M = a.M;
                if (a.volume) {                          // model.py:124-128, fp64 then one rounding to fp32
                    const uint32_t k = (uint32_t)(p % a.D), j = (uint32_t)((p / a.D) % a.W), i = (uint32_t)(p / ((int64_t)a.D * a.W));
                    c[0] = (float)(((double)i / (double)(a.H - 1)) * 2.0 - 1.0);
                    c[1] = (float)(((double)j / (double)(a.W - 1)) * 2.0 - 1.0);
                    c[2] = (float)(((double)k / (double)(a.D - 1)) * 2.0 - 1.0);
                    const size_t hwd = (size_t)a.H * a.W * a.D;
#pragma unroll
                    for (uint32_t g = 0; g < kMaxMods; ++g) m[g] = M ? a.feats[(size_t)(g < M ? g : M - 1) * hwd + p] : 0.0f;
                } else {
                    c[0] = a.coords[p * 3 + 0]; c[1] = a.coords[p * 3 + 1]; c[2] = a.coords[p * 3 + 2];
#pragma unroll
                    for (uint32_t g = 0; g < kMaxMods; ++g) m[g] = M ? a.feats[p * (int64_t)M + (g < M ? g : M - 1)] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) raw[32 * k] = c[k];
#pragma unroll
                for (int k = 0; k < kMaxMods; ++k) raw[32 * (3 + k)] = m[k];
                raw[32 * (kRawStride - 1)] = 0.0f;
            }
            // ... and both halves build their features from the table (same wave: LDS ops are in order)
            // (eight descriptors, then eight raw values, then the arithmetic: two LDS latencies per k step
            // instead of two per feature)
#pragma unroll
            for (int t = 0; t < KT0; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float4 d[8];
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = ldsTab[32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = raw[32 * __builtin_bit_cast(uint32_t, d[j].z)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t mode = __builtin_bit_cast(uint32_t, d[j].w);
                        const float v = mode == 1u ? __builtin_amdgcn_sinf(__builtin_fmaf(x[j], d[j].x, d[j].y)) : x[j];
                        const __bf16 hi = (__bf16)v;
                        const __bf16 lo = (__bf16)(v - (float)hi);
                        xin_hi[t][s][j] = (AUG && mode == 2u) ? lo : hi;
                        xin_lo[t][s][j] = lo;
                    }
                }
        } else {
            // raw-x kinds (tests, other front ends): feats IS the [n][inDim] input matrix
#pragma unroll
            for (int t = 0; t < KT0; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t f = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                        v[j] = a.feats[p * (int64_t)a.L.inDim + (f < a.L.inDim ? f : 0u)];
                        v[j] = f < a.L.inDim ? v[j] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const __bf16 hi = (__bf16)v[j];
                        xin_hi[t][s][j] = hi;
                        xin_lo[t][s][j] = (__bf16)(v[j] - (float)hi);
                    }
                }
        }
    };

    bf16x8 Hc[KT][2];            // current layer's input, as B operands
    bf16x8 Hn[KT][2];            // next layer's input, built out tile by out tile

    // bias of out tile o for this lane half: rows 8g + 4h .. +3 (g = 0..3) = four aligned float4
    // The accumulator STARTS at the bias (accumulator register i holds row (i&3) + 8(i>>2) + 4h, i.e. the
    // four aligned float4 at rows 8g + 4h of the LDS copy), so no add is left for the activation.
    auto bias_tile = [&](uint32_t layerOff, int o) {                     // ds_read_b128 x 4
        f32x16 acc;
#ifdef MRIRT_EXP_NOBIAS
        return (f32x16)(0.0f);
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = ldsBias[(layerOff + 32 * o + 8 * g + 4 * h) >> 2];
            acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
        }
        return acc;
    };
    // one instruction per value (v_sin_f32 on revolutions: the 1/2pi and w0 are folded into the weights;
    // neumors_inr.ipynb:1165-1178) or half of one (v_pk_max_f32; model.py:46-48), plus the packed bf16 convert
    auto activate = [&](const f32x16& acc, int o) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int i = 8 * s + j;
                f32x2 x = { acc[i], acc[i + 1] };
#ifdef MRIRT_EXP_NOSIN
                if (siren) { }
#else
                if (siren) { x.x = __builtin_amdgcn_sinf(x.x); x.y = __builtin_amdgcn_sinf(x.y); }
#endif
                else x = __builtin_elementwise_max(x, (f32x2){ 0.0f, 0.0f });
                Hn[o][s][j] = (__bf16)x.x;
                Hn[o][s][j + 1] = (__bf16)x.y;
            }
    };
    // The same activation, one slice per k step of the NEXT out tile.  The two waves of a SIMD leave the
    // chunk barrier in phase; with whole-tile activation they would both be in their VALU phase at once
    // and the matrix core would idle (measured: MFMA-busy + VALU cycles == kernel cycles).  Sliced under
    // the next tile's MFMAs, a v_sin_f32 (16 cycles) fits inside each MFMA's 32.
    auto act_slice = [&](f32x16& ap, int o, int step, auto nstepsC) {
        constexpr int NS = decltype(nstepsC)::value, PER = 16 / NS;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = step * PER + q;
#ifndef MRIRT_EXP_NOSIN
            if (siren) ap[i] = __builtin_amdgcn_sinf(ap[i]);
#endif
            if (i & 1) {
                f32x2 x = { ap[i - 1], ap[i] };
                if (!siren) x = __builtin_elementwise_max(x, (f32x2){ 0.0f, 0.0f });
                Hn[o][i >> 3][(i & 7) - 1] = (__bf16)x.x;
                Hn[o][i >> 3][i & 7] = (__bf16)x.y;
            }
        }
    };
    // the chunk prefetched during this chunk's compute becomes current (hipcc drains the LDS-DMA
    // with vmcnt(0) at the barrier)
    auto next_chunk = [&]() {
        if constexpr (RES) curFrag = nextFrag;           // nothing is staged, nothing to wait for
#ifdef MRIRT_EXP_NOBARRIER
        else { buf ^= 1; }
#else
        else { __syncthreads(); buf ^= 1; }
#endif
    };

    __syncthreads();                                     // biases and layer-0 chunk 0 are in LDS

    // Persistent workgroup: batches of 256 points, round-robin.  The weight stream is cyclic — the head
    // chunk stages layer 0's first chunk of the NEXT batch — so after the first batch no DMA latency,
    // workgroup launch or bias staging is exposed.
    const int64_t nBatches = (nPts + kInrWaves * 32 - 1) / (kInrWaves * 32);
    int64_t pendIdx = -1;
    int16_t pendVal = 0;
    for (int64_t batch = blockIdx.x; batch < nBatches; batch += gridDim.x) {
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    r = lane & 31u; h = lane >> 5;
    if (pendIdx >= 0) a.argmax[pendIdx] = pendVal;       // the previous batch's classes
    const int64_t pidx = (batch * kInrWaves + waveS) * 32 + r;
    load_inputs(pidx);

    // ---- layer 0: SIREN: hi/lo split, three products per k step; ReLU nets: one ------------------------------
    {
        const uint32_t b0 = a.L.biasOff[0];
#pragma unroll
        for (int og = 0; og < KT / OTC; ++og) {
            const int nfragNext = (og + 1 < KT / OTC) ? CH0 : (a.L.numLayers > 2 ? CHH : FH);   // layer 0 / layer 1 / head
            bf16x8 ring[RD];                             // fragment stream (SPLIT: even = hi, odd = lo of k step f/2)
#pragma unroll
            for (int d = 0; d < RD; ++d) if (d < CH0) ring[d] = frag_at(buf, d);
#pragma unroll
            for (int oo = 0; oo < OTC; ++oo) {
                const int o = og * OTC + oo;
                f32x16 acc = bias_tile(b0, o);
#pragma unroll
                for (int p = 0; p < KT0 * 2; ++p) {
                    const int t = p >> 1, s = p & 1, f = oo * F0 + (SPLIT ? 2 : 1) * p, step = oo * KT0 * 2 + p;
                    if (step % SP0 == 0 && step / SP0 < PERW) {
                        if (og + 1 < KT / OTC) stage_piece(nextFrag, IC<CH0>{}, buf ^ 1, step / SP0);    // folds: og is unrolled
                        else                   stage_piece(nextFrag, IC<CHH>{}, buf ^ 1, step / SP0);
                    }
                    if constexpr (SPLIT) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(f + 1) % RD], xin_hi[t][s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f % RD], xin_lo[t][s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f % RD], xin_hi[t][s], acc, 0, 0, 0);
                        if (f + RD < CH0) ring[f % RD] = frag_at(buf, f + RD);
                        if (f + 1 + RD < CH0) ring[(f + 1) % RD] = frag_at(buf, f + 1 + RD);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f % RD], xin_hi[t][s], acc, 0, 0, 0);
                        if (f + RD < CH0) ring[f % RD] = frag_at(buf, f + RD);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                activate(acc, o);
            }
            next_chunk();
            nextFrag += nfragNext;
        }
    }

    // ---- hidden layers 1 .. L-2 ----------------------------------------------------------------------------
    for (uint32_t l = 1; l + 1 < a.L.numLayers; ++l) {
#pragma unroll
        for (int t = 0; t < KT; ++t) { Hc[t][0] = Hn[t][0]; Hc[t][1] = Hn[t][1]; }
        const uint32_t bl = a.L.biasOff[l];
        const bool lastHidden = l + 2 == a.L.numLayers;
        f32x16 accPrev;                                  // finished tile o - 1, activated under tile o's MFMAs
#pragma unroll
        for (int og = 0; og < KT / OTC; ++og) {
            const int nfragNext = (og + 1 < KT / OTC || !lastHidden) ? CHH : FH;    // more hidden tiles, or the head
            // A fragments run RD ahead of the MFMA that consumes them (the chunk is one fragment stream,
            // so the prefetch carries across out tiles): an LDS read is ~4 MFMA issue slots of latency
            bf16x8 ring[RD];
#pragma unroll
            for (int d = 0; d < RD; ++d) if (d < CHH) ring[d] = frag_at(buf, d);
#pragma unroll
            for (int oo = 0; oo < OTC; ++oo) {
                const int o = og * OTC + oo;
                f32x16 acc = bias_tile(bl, o);
#pragma unroll
                for (int ts = 0; ts < FH; ++ts) {
                    const int f = oo * FH + ts;
                    // right after the tile's bias read: hipcc makes any LDS read it cannot tell apart from the
                    // DMA target wait vmcnt(0), so a piece still in flight at the next tile's bias read stalls it
                    if constexpr (FH > PPT) {
                        if (ts >= 1 && ts <= PPT && oo * PPT + ts - 1 < PERW) stage_piece(nextFrag, IC<CHH>{}, buf ^ 1, oo * PPT + ts - 1);
                    } else {
                        if (f % SPH == 0 && f / SPH < PERW) stage_piece(nextFrag, IC<CHH>{}, buf ^ 1, f / SPH);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f % RD], Hc[ts >> 1][ts & 1], acc, 0, 0, 0);
                    if (f + RD < CHH) ring[f % RD] = frag_at(buf, f + RD);
                    if (o > 0) act_slice(accPrev, o - 1, ts, IC<FH>{});
                    __builtin_amdgcn_sched_barrier(0);   // keep the refill RD steps ahead (the scheduler would sink it)
                }
                accPrev = acc;
            }
            next_chunk();
            nextFrag += nfragNext;
        }
#pragma unroll
        for (int ts = 0; ts < FH; ++ts) act_slice(accPrev, KT - 1, ts, IC<FH>{});     // the layer's last tile
    }

    // ---- head: one out tile (outDim <= 16 rows used), linear ---------------------------------------------------
    {
        f32x16 acc = bias_tile(a.L.biasOff[a.L.numLayers - 1], 0);      // rows >= outDim: zero padding
        stage_issue(a.L.fragOff[0], IC<CH0>{}, buf ^ 1);                 // the next batch's first chunk
        bf16x8 ring[RD];
#pragma unroll
        for (int d = 0; d < RD; ++d) if (d < FH) ring[d] = frag_at(buf, d);
#pragma unroll
        for (int f = 0; f < FH; ++f) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[f % RD], Hn[f >> 1][f & 1], acc, 0, 0, 0);
            if (f + RD < FH) ring[f % RD] = frag_at(buf, f + RD);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.logits != nullptr && pidx < nPts) {         // uniform pointer test; not the throughput path
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t cls = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (cls < a.L.outDim) a.logits[pidx * a.L.outDim + cls] = acc[i];
            }
        }
        // np.argmax = first maximum.  Within a lane cls grows with i, so a strict > keeps the first; the
        // two lane halves hold interleaved classes, so the merge breaks ties towards the smaller class.
        float best = -INFINITY, second = -INFINITY;
        uint32_t bestc = 4 * h;                          // this half's first class (all -inf: class 0, as numpy)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t cls = (i & 3) + 8 * (i >> 2) + 4 * h;
            const float v = cls < a.L.outDim ? acc[i] : -INFINITY;
            top2_push(v, cls, best, second, bestc);
        }
        const float ob = __shfl_xor(best, 32), os = __shfl_xor(second, 32);
        const uint32_t oc = __shfl_xor(bestc, 32);
        if (ob > best || (ob == best && oc < bestc)) { second = fmaxf(best, os); best = ob; bestc = oc; }
        else second = fmaxf(second, ob);
        // near-tie mark (see kFlagBit): the refinement kernel re-evaluates this point; NaN gaps are marked too
        const bool tie = a.tie != nullptr && !(best - second >= tieThr);
        // stored at the top of the next batch: issued here, the store would still be in flight at the
        // chunk barrier below, whose vmcnt(0) (for the DMA) would then wait out its whole HBM round trip
        pendIdx = (a.argmax && h == 0 && pidx < nPts) ? pidx : -1;
        pendVal = (int16_t)(bestc | (tie ? (uint32_t)kFlagBit : 0u));
    }
    next_chunk();
    nextFrag = a.L.fragOff[0] + CH0;
    if constexpr (RES) curFrag = a.L.fragOff[0];
    }   // batch
    if (pendIdx >= 0) a.argmax[pendIdx] = pendVal;
}

// =====================================================================================================
// Weight-stationary kernel for BASELINE config 5's network: the 4 x 256 SIREN on (coords, 4 modalities),
// 7 -> 256 -> 256 -> 256 -> 256 -> (<= 4 classes)   (notebooks/neumors_inr.ipynb:853-899,1165-1178).
//
// Why a second dataflow.  Measured on the streaming kernel above (profiles/r02_inr_knockouts.txt): with the
// weight LDS-DMA compiled out it runs 24.7 -> 18.8 ms, with the input staging compiled out 24.7 -> 20.1 ms.  A SIMD
// has 32 issue cycles per MFMA; the stream spends ~12 of them per MFMA on DMA pieces (416 KiB of weights pass
// through LDS for every 256 points, and the point count per CU is capped by the register file that holds the
// activations).  Here the roles are swapped: the WEIGHTS live in the register file for the whole launch — a
// CU's four SIMDs hold 4 x 128 KiB of registers, the network is 416 KiB — and the ACTIVATIONS travel through
// LDS.  One wave per SIMD (512 registers); wave w owns out tiles 2w and 2w+1 of the three hidden layers (384 registers
// of A fragments: layers 1 and 2 in the AGPR half through "a"-constrained MFMAs, layer 3 in VGPRs, its last four
// fragments and the small layer-0 / head fragments in LDS); a round is G = 3 groups of 32 points; per layer a wave reads
// a group's sixteen 1-KiB B fragments from LDS (what all four waves wrote in the previous phase), runs its
// 2 x 16 MFMAs, and writes its two out tiles back as the next layer's B fragments 4w .. 4w+3 (the accumulator
// layout IS the B layout, as above: two ds_write_b128 per tile).  The head is one out tile: wave g (g < G) runs it
// for group g, one round late, interleaved with the next round's layer 0 (which is VALU-bound: two MFMAs and sixteen
// sines per tile pass, while the head is sixteen MFMAs and no activation).  No weight traffic at all after the
// prologue, one workgroup barrier per phase (four per round).
// Tried and measured no faster (round 2, s_memtime stamps + wall): a second schedule that chains the passes across phase
// and round boundaries with two write-only barriers per phase (post-op always under the next pass, ring running on
// into the next layer) — 20.0 k cycles per round against 18.1 k here; a micro-benchmark of the pass (tools/micro/
// mfma_pace.hip) shows why neither wins: 16 MFMAs + 16 LDS reads take 549 cycles, and the 16 v_sin + 8 converts of the
// previous tile's activation add ~160 on top instead of hiding (709) — with one wave per SIMD the VALU work of the gaps
// that carry two sines overflows the 24 issue cycles an MFMA leaves.
// Same packed image, same arithmetic per point (accumulator starts at the bias, k steps in ascending order):
// bit-identical logits to the streaming kernel (tools/ws_check.py; the tests hold both kernels to the same fp32 /
// fp64 references, and each is batch-position invariant by construction).
// =====================================================================================================
constexpr int kWsGroups = 3;
// Diagnostic build (-DMRIRT_WS_STAMPS): s_memtime at the phase boundaries of one steady-state round of every
// workgroup goes to the logits buffer (which then holds no logits): [block][wave][24] uint64.  Never in the product.
#ifdef MRIRT_WS_STAMPS
#define WS_STAMP(k) do { if (stampRound && lane == 0) stamps[((size_t)blockIdx.x * 4 + w) * 24 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(k) do { } while (0)
#endif
#ifdef MRIRT_WS_NOBAR
#define WS_SYNC() do { } while (0)
#else
#define WS_SYNC() __syncthreads()
#endif

// MFMA with the operand register class spelled out.  hipcc's allocator fills the 256 arch VGPRs with weights first
// and then shuttles everything VALU touches through v_accvgpr moves (measured: 220 VGPRs of weights, 520
// v_accvgpr_read per round, 21 fragments spilled to scratch and reloaded every round).  An "a" constraint pins a
// fragment into the accumulation half of the unified file — which VALU cannot use anyway — and leaves the arch
// half to the accumulators, the B ring and the activation.  The compiler does not know these are matrix
// instructions, so the MFMA -> VALU read hazard is kept by construction (see the call sites) or by mfma_drain(acc).
__device__ __forceinline__ void mfma_a(f32x16& acc, const bf16x8& wa, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& acc, const bf16x8& wv, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wv), "v"(b));
}
// after the last MFMA of a chain whose result VALU reads next: 8 passes + 3 wait states, taken generously
// (the accumulator is an in/out operand so that no read of it can be scheduled above the wait)
__device__ __forceinline__ void mfma_drain(f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc)); }

template <bool SIREN>
__global__ __launch_bounds__(256, 1) void inr_ws_kernel(const InrArgs a) {
#ifndef MRIRT_WS_RD
#define MRIRT_WS_RD 4
#endif
    constexpr int G = kWsGroups, NH = 3, KS = 16, RD = MRIRT_WS_RD;      // RD: B fragments in flight ahead of their MFMA
    constexpr int kActQ = G * 2 * KS * 64;               // uint4: [group][parity][k step][lane]    96 KiB
    constexpr int kInQ = G * 2 * 64;                     // layer-0 B fragments (k steps 0, 1)       6 KiB
    constexpr int kPartQ = 0;                            // (head partials of an earlier k-split head: gone)
    constexpr int kBiasQ = (4 * 256 + 32) / 4;           // padded biases                          4.1 KiB
    constexpr int kW0Q = 16 * 64, kWoQ = 16 * 64;        // layer-0 and head A fragments (all waves')   16 + 16 KiB
    constexpr int kW3Q = 4 * 4 * 64;                     // four layer-3 A fragments per wave that do not fit the registers  16 KiB
    constexpr int kW0Off = kActQ + kInQ + kPartQ + kBiasQ, kWoOff = kW0Off + kW0Q, kW3Off = kWoOff + kWoQ;
    __shared__ uint4 ldsAll[kActQ + kInQ + kPartQ + kBiasQ + kW0Q + kWoQ + kW3Q];

    int64_t nPts = a.n;
    if (a.nDev != nullptr) { const int64_t nd = (int64_t)*a.nDev; nPts = nd < nPts ? nd : nPts; }
    if (nPts <= 0) return;
    const float tieThr = a.tie != nullptr ? a.tie[0] * a.tieScale : 0.0f;

    const uint32_t lane = threadIdx.x & 63u, r = lane & 31u, h = lane >> 5;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // biases -> LDS once (scaled like the packed weights: see InrLayout::bscale)
        const uint32_t nq = (a.L.biasOff[a.L.numLayers - 1] + 32) / 4;
        for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) {
            uint32_t l = 0;
            while (l + 1 < a.L.numLayers && 4 * i >= a.L.biasOff[l + 1]) ++l;
            float4 b = reinterpret_cast<const float4*>(a.bias)[i];
            const float sc = a.L.bscale[l];
            b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
            ldsAll[kActQ + kInQ + kPartQ + i] = __builtin_bit_cast(uint4, b);
        }
    }
    // ---- the network, resident in this wave's registers --------------------------------------------------
    const uint4* __restrict__ wp = a.wpack;
    bf16x8 Wh[NH][2][KS];                                 // 384 registers: the three hidden matrices' out tiles 2w, 2w+1
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 fr = wp[(size_t)(a.L.fragOff[l + 1] + (2 * w + j) * KS + ks) * 64 + lane];
                // 96 fragments = 384 registers is what fits beside the accumulators: the last four of layer 3 go through LDS
                if (l == NH - 1 && j == 1 && ks >= KS - 4) ldsAll[kW3Off + (w * 4 + (ks - (KS - 4))) * 64 + lane] = fr;
                else Wh[l][j][ks] = __builtin_bit_cast(bf16x8, fr);
            }
    // layer 0 (2 k steps per out tile) and the head (k-quarter per wave) are 4 + 4 fragments per wave: they stay in
    // LDS (lane-linear, read just before use) — 400 + of the 512 registers as weights left the allocator no room
    for (uint32_t f = w; f < 16; f += 4) {
        ldsAll[kW0Off + f * 64 + lane] = wp[(size_t)(a.L.fragOff[0] + f) * 64 + lane];          // [tile o][s] = fragment 2o + s
        ldsAll[kWoOff + f * 64 + lane] = wp[(size_t)(a.L.fragOff[NH + 1] + f) * 64 + lane];     // head k step f
    }
    // LDS addressing: a handful of per-lane base pointers (re-laundered every round so that the compiler cannot
    // expand them into one hoisted VGPR per fragment address and spill) + compile-time immediates (<= 64 KiB each).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));       // plain vector types: HIP's uint4 / float4 classes
    typedef float f32x4 __attribute__((ext_vector_type(4)));          // cannot be assigned through an LDS-qualified pointer
    typedef __attribute__((address_space(3))) u32x4 lds_q;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_q* bR = (lds_q*)ldsAll + lane;                               // B-fragment reads: fragment f at bR[64 f]
    lds_q* bW = (lds_q*)ldsAll + 4 * w * 64 + lane;                  // this wave's output fragments 4w .. 4w+3
    lds_q* bIn = (lds_q*)ldsAll + kActQ + lane;                      // layer-0 B fragments
    lds_f4* bPart = (lds_f4*)((lds_q*)ldsAll + kActQ + kInQ) + w * 32 + r;
    lds_f4* bBias = (lds_f4*)((lds_q*)ldsAll + kActQ + kInQ + kPartQ) + 16 * w + h;      // tile 2w, rows 4h..
    lds_q* bWt = (lds_q*)ldsAll + kW0Off + 4 * w * 64 + lane;        // this wave's layer-0 fragments 4w..4w+3; head: + kW0Q;
                                                                     // its four LDS-held layer-3 fragments: + kW0Q + kWoQ
    auto launder = [&]() {
        uint32_t v0 = (uint32_t)(uintptr_t)bR, v1 = (uint32_t)(uintptr_t)bW, v2 = (uint32_t)(uintptr_t)bIn,
                 v3 = (uint32_t)(uintptr_t)bPart, v4 = (uint32_t)(uintptr_t)bBias, v5 = (uint32_t)(uintptr_t)bWt;
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
        bWt = (lds_q*)(uintptr_t)v5;
        bR = (lds_q*)(uintptr_t)v0; bW = (lds_q*)(uintptr_t)v1; bIn = (lds_q*)(uintptr_t)v2;
        bPart = (lds_f4*)(uintptr_t)v3; bBias = (lds_f4*)(uintptr_t)v4;
    };
    // biases of this specialised network sit at layer l -> 256 l (head: 1024): compile-time immediates
    auto bias_tile = [&](int l, int j) {                                 // tile 2w + j: rows 8g + 4h .. +3 (g = 0..3)
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = bBias[(256 * l + 32 * j + 8 * g) >> 2];
            acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
        }
        return acc;
    };
    auto rd_frag = [&](int g, int par, int ks) { return __builtin_bit_cast(bf16x8, bR[((g * 2 + par) * KS + ks) * 64]); };
    auto wr_frag = [&](int g, int par, int j, int s2, const bf16x8& v) { bW[((g * 2 + par) * KS + 2 * j + s2) * 64] = __builtin_bit_cast(u32x4, v); };

    // raw inputs of point `pidx` -> the two layer-0 B fragments of the augmented split (see make_layout):
    //   k step 0 = [x_hi(0..6), x_lo(0..6), 0, 0],  k step 1 = [x_hi(0..6), 0 ...]; element j of lane half h is slot 8(j>>2) + 4h + (j&3)
    float rawc[3];
    float4 rawf;
    auto load_raw = [&](int64_t pidx) {
        const int64_t p = pidx < nPts ? pidx : nPts - 1;
        rawc[0] = a.coords[p * 3 + 0]; rawc[1] = a.coords[p * 3 + 1]; rawc[2] = a.coords[p * 3 + 2];
        rawf = reinterpret_cast<const float4*>(a.feats)[p];
    };
    auto stage_raw = [&](int g) {
        const float x[7] = { rawc[0], rawc[1], rawc[2], rawf.x, rawf.y, rawf.z, rawf.w };
        __bf16 hi[8], lo[8];
#pragma unroll
        for (int k = 0; k < 7; ++k) { hi[k] = (__bf16)x[k]; lo[k] = (__bf16)(x[k] - (float)hi[k]); }
        hi[7] = (__bf16)0.0f; lo[7] = (__bf16)0.0f;
        const __bf16 z = (__bf16)0.0f;
        bf16x8 f0, f1;
        // lane half 0: slots 0..3, 8..11        lane half 1: slots 4..7, 12..15
        const bf16x8 f0a = { hi[0], hi[1], hi[2], hi[3], lo[1], lo[2], lo[3], lo[4] };
        const bf16x8 f0b = { hi[4], hi[5], hi[6], lo[0], lo[5], lo[6], z, z };
        const bf16x8 f1a = { hi[0], hi[1], hi[2], hi[3], z, z, z, z };
        const bf16x8 f1b = { hi[4], hi[5], hi[6], z, z, z, z, z };
        f0 = h ? f0b : f0a;
        f1 = h ? f1b : f1a;
        lds_q* dst = (lds_q*)ldsAll + kActQ + g * 2 * 64 + lane;      // g = this wave's id: runtime, once per round
        dst[0] = __builtin_bit_cast(u32x4, f0);
        dst[64] = __builtin_bit_cast(u32x4, f1);
    };

    // one value of a finished accumulator per call: activation, and every second value the packed bf16 convert
    auto act_one = [&](f32x16& ap, bf16x8 (&Ho)[2], int i) {
#ifndef MRIRT_WS_NOSIN
        if constexpr (SIREN) ap[i] = __builtin_amdgcn_sinf(ap[i]);
#endif
        if (i & 1) {
            f32x2 x = { ap[i - 1], ap[i] };
            if constexpr (!SIREN) x = __builtin_elementwise_max(x, (f32x2){ 0.0f, 0.0f });
            Ho[i >> 3][(i & 7) - 1] = (__bf16)x.x;
            Ho[i >> 3][i & 7] = (__bf16)x.y;
        }
    };

    // The same in two sweeps, for the places where no MFMA stream separates a sine from the convert that reads it (layer 0's passes,
    // the last hidden layer's tail): sixteen independent transcendentals first, then the eight packed converts — a convert right
    // behind its sines waits out the transcendental's latency, and with one wave per SIMD nothing else can issue meanwhile
    // (s_memtime stamps: ~70 cycles per pass).
    auto sin_one = [&](f32x16& ap, int i) {
#ifndef MRIRT_WS_NOSIN
        if constexpr (SIREN) ap[i] = __builtin_amdgcn_sinf(ap[i]);
#endif
    };
    auto cvt_all = [&](f32x16& ap, bf16x8 (&Ho)[2]) {
#pragma unroll
        for (int i = 1; i < 16; i += 2) {
            f32x2 x = { ap[i - 1], ap[i] };
            if constexpr (!SIREN) x = __builtin_elementwise_max(x, (f32x2){ 0.0f, 0.0f });
            Ho[i >> 3][(i & 7) - 1] = (__bf16)x.x;
            Ho[i >> 3][i & 7] = (__bf16)x.y;
        }
    };

    // Head (linear, <= 4 classes): one out tile, so one wave per group — wave g reads group g's last hidden outputs
    // (parity NH & 1 ^ 1 ... written by all four waves before the barrier that precedes this) and the sixteen head
    // fragments from LDS, and stores logits / argmax.  np.argmax = first maximum.
    auto head = [&](int64_t rnd) {
        constexpr int par = (NH & 1) ^ 1;
        f32x16 acc;
        {
            const f32x4 hb = *((lds_f4*)((lds_q*)ldsAll + kActQ + kInQ + kPartQ) + (1024 >> 2) + h);   // rows 4h..4h+3 of the head's bias
            acc = (f32x16)(0.0f);
            acc[0] = hb.x; acc[1] = hb.y; acc[2] = hb.z; acc[3] = hb.w;
        }
        lds_q* bH = (lds_q*)ldsAll + (w * 2 + par) * KS * 64 + lane;
        lds_q* bWo = (lds_q*)ldsAll + kWoOff + lane;
        asm volatile("s_nop 3" : "+v"(acc));                             // VALU wrote acc (v_mov): two wait states before an MFMA reads it as C
        bf16x8 ra[4], rb[4];                                              // four k steps of operands in flight
#pragma unroll
        for (int d = 0; d < 4; ++d) { ra[d] = __builtin_bit_cast(bf16x8, bWo[d * 64]); rb[d] = __builtin_bit_cast(bf16x8, bH[d * 64]); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            mfma_v(acc, ra[ks % 4], rb[ks % 4]);
            if (ks + 4 < KS) { ra[ks % 4] = __builtin_bit_cast(bf16x8, bWo[(ks + 4) * 64]); rb[ks % 4] = __builtin_bit_cast(bf16x8, bH[(ks + 4) * 64]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_drain(acc);
        const int64_t pidx = (rnd * G + w) * 32 + r;
        if (h == 0 && pidx < nPts) {                                      // classes 0..3 are rows 0..3: lane half 0, registers 0..3
            const float v[4] = { acc[0], acc[1], acc[2], acc[3] };
#ifndef MRIRT_WS_STAMPS
            if (a.logits != nullptr)
                for (uint32_t c = 0; c < a.L.outDim; ++c) a.logits[pidx * a.L.outDim + c] = v[c];
#endif
            if (a.argmax != nullptr) {
                float best = -INFINITY, second = -INFINITY;
                uint32_t bestc = 0;
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) top2_push(c < a.L.outDim ? v[c] : -INFINITY, c, best, second, bestc);   // strict: the first maximum
                const bool tie = a.tie != nullptr && !(best - second >= tieThr);     // near-tie mark (kFlagBit)
                a.argmax[pidx] = (int16_t)(bestc | (tie ? (uint32_t)kFlagBit : 0u));
            }
        }
    };

    const int64_t nRounds = (nPts + G * 32 - 1) / (G * 32);
    int64_t round = blockIdx.x;
    if (w < G) {                                           // wave w stages (and later reduces) group w
        load_raw((round * G + w) * 32 + r);
        stage_raw((int)w);
    }
    __syncthreads();                                       // biases + first inputs are in LDS
    int64_t prevRound = -1;

    for (; round < nRounds; round += gridDim.x) {
        const int64_t nextRound = round + gridDim.x;
        launder();
#ifdef MRIRT_WS_STAMPS
        uint64_t* stamps = reinterpret_cast<uint64_t*>(a.logits);
        const bool stampRound = stamps != nullptr && round == (int64_t)blockIdx.x + 3 * (int64_t)gridDim.x;
#endif
        WS_STAMP(0);
        // ---- layer 0 of this round + head of the previous one, interleaved ------------------------------------
        // Layer 0 is VALU-bound (2 MFMAs but 16 sines per tile pass), the head is MFMA-bound (16 MFMAs, no activation),
        // and run one after the other they cost a sixth of a round (profiles/r02_inr_ws_knockouts.txt).  Here pass
        // i's activation is sliced under pass i+1's MFMAs and two or three head MFMAs (waves 0..G-1: wave g takes
        // group g of the PREVIOUS round, sixteen k steps over the last hidden layer's outputs) per pass.
        f32x16 carry;                                      // a phase's last accumulator, activated under the next phase's pass 0
        // The head's epilogue (the previous round's logits -> stores, argmax, near-tie mark): ~70 VALU instructions, ~560 cycles on
        // waves 0 .. G-1 at the end of the layer-0 phase (s_memtime stamps).  Moving it behind the barrier, under hidden layer 1's
        // first LDS reads, moves the same 560 cycles there (measured: cycles per round level, wall time +0.3 %): it stays here.
        f32x16 headAcc;
        const bool headPending = prevRound >= 0 && w < G;
        auto head_epilogue = [&]() {
            if (!headPending) return;
            mfma_drain(headAcc);
            const int64_t pidx = (prevRound * G + w) * 32 + r;
            if (h == 0 && pidx < nPts) {                       // classes 0..3 are rows 0..3: lane half 0, registers 0..3
                const float v[4] = { headAcc[0], headAcc[1], headAcc[2], headAcc[3] };
#ifndef MRIRT_WS_STAMPS
                if (a.logits != nullptr)
                    for (uint32_t c = 0; c < a.L.outDim; ++c) a.logits[pidx * a.L.outDim + c] = v[c];
#endif
                if (a.argmax != nullptr) {
                    float best = -INFINITY, second = -INFINITY;
                    uint32_t bestc = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) top2_push(c < a.L.outDim ? v[c] : -INFINITY, c, best, second, bestc);   // strict: the first maximum
                    const bool tie = a.tie != nullptr && !(best - second >= tieThr);     // near-tie mark (kFlagBit)
                    a.argmax[pidx] = (int16_t)(bestc | (tie ? (uint32_t)kFlagBit : 0u));
                }
            }
        };
#if !defined(MRIRT_WS_NOL0)
        auto l0_head = [&](auto headC) {
            constexpr bool HEAD = decltype(headC)::value;
            constexpr int hpar = (NH & 1) ^ 1;                // parity the last hidden layer wrote
            lds_q* bH = (lds_q*)ldsAll + (w * 2 + hpar) * KS * 64 + lane;
            lds_q* bWo = (lds_q*)ldsAll + kWoOff + lane;
            f32x16 hacc;
            bf16x8 ra[2], rb[2];                               // head operands, two k steps ahead
            if constexpr (HEAD) {
                const f32x4 hb = *((lds_f4*)((lds_q*)ldsAll + kActQ + kInQ + kPartQ) + (1024 >> 2) + h);   // rows 4h..4h+3 of the head's bias
                hacc = (f32x16)(0.0f);
                hacc[0] = hb.x; hacc[1] = hb.y; hacc[2] = hb.z; hacc[3] = hb.w;
#pragma unroll
                for (int d = 0; d < 2; ++d) { ra[d] = __builtin_bit_cast(bf16x8, bWo[d * 64]); rb[d] = __builtin_bit_cast(bf16x8, bH[d * 64]); }
                asm volatile("s_nop 3" : "+v"(hacc));          // VALU wrote hacc (v_mov): wait states before an MFMA reads it as C
            }
            int hk = 0;                                        // head k step (a compile-time value in the unrolled code)
            auto head_step = [&]() {
                if constexpr (HEAD) {
                    if (hk < KS) {
                        mfma_v(hacc, ra[hk % 2], rb[hk % 2]);
                        if (hk + 2 < KS) { ra[hk % 2] = __builtin_bit_cast(bf16x8, bWo[(hk + 2) * 64]); rb[hk % 2] = __builtin_bit_cast(bf16x8, bH[(hk + 2) * 64]); }
                        ++hk;
                    }
                }
            };
            constexpr int NP = 2 * G;
            f32x16 acc = bias_tile(0, 0), accPrev, accNext;
            bf16x8 Ho[2];
            // the four layer-0 A fragments of this wave's two tiles are read ONCE per phase, the input fragments one group ahead:
            // read pass by pass right in front of their MFMAs, every pass exposed an LDS round trip (round 2's stamps: 3.6 k
            // cycles for a phase whose MFMA + VALU work is ~2 k).
            bf16x8 wa[4], Bin[2], BinNext[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) wa[q] = __builtin_bit_cast(bf16x8, bWt[q * 64]);
            Bin[0] = __builtin_bit_cast(bf16x8, bIn[0]); Bin[1] = __builtin_bit_cast(bf16x8, bIn[64]);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int g = i >> 1, j = i & 1;
                WS_STAMP(15 + i);
                if (j == 0 && g + 1 < G) { BinNext[0] = __builtin_bit_cast(bf16x8, bIn[((g + 1) * 2 + 0) * 64]); BinNext[1] = __builtin_bit_cast(bf16x8, bIn[((g + 1) * 2 + 1) * 64]); }
                mfma_v(acc, wa[2 * j + 0], Bin[0]);
                mfma_v(acc, wa[2 * j + 1], Bin[1]);
                if (j == 1) { Bin[0] = BinNext[0]; Bin[1] = BinNext[1]; }
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < NP) accNext = bias_tile(0, (i + 1) & 1);
                if (i > 0) {
                    const int pg = (i - 1) >> 1, pj = (i - 1) & 1;
                    asm volatile("s_nop 15" : "+v"(accPrev));   // two MFMA issues + 16 states since accPrev's producer
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        sin_one(accPrev, v);
                        if (v == 3 || v == 9 || (v == 14 && i < 5)) { head_step(); __builtin_amdgcn_sched_barrier(0); }
                    }
                    cvt_all(accPrev, Ho);
                    wr_frag(pg, 1, pj, 0, Ho[0]);
                    wr_frag(pg, 1, pj, 1, Ho[1]);
                } else {
                    head_step(); head_step(); head_step();
                }
                __builtin_amdgcn_sched_barrier(0);
                accPrev = acc;
                if (i + 1 < NP) acc = accNext;
            }
            // the last pass's activation is carried into hidden layer 1's pass 0 (see `carry` below); what is left of the head:
            WS_STAMP(21);
            head_step();
            carry = accPrev;
            if constexpr (HEAD) {
                while (hk < KS) head_step();                   // (none left with G = 3: 3 + 3*4 ... counted at compile time)
                headAcc = hacc;
                head_epilogue();
            }
        };
#ifdef MRIRT_WS_NOHEAD
        l0_head(IC<0>{});
#else
        if (prevRound >= 0 && w < G) l0_head(IC<1>{}); else l0_head(IC<0>{});
#endif
#endif
        WS_STAMP(1);
        WS_SYNC();
        WS_STAMP(2);
        // ---- hidden layers: layer l reads parity (l & 1), writes the other; the last one feeds the head ---------
        // The last pass of a layer has no MFMAs of its own layer left to hide its activation under: as a tail after the loop it
        // cost 476 cycles per layer (s_memtime stamps, profiles/r04_inr_ws/).  Layers 0, 1 and 2 therefore hand their last
        // accumulator to the NEXT layer (`carry`), whose pass 0 — which had nothing to cover — activates it and writes it into the
        // buffer that layer reads (group G - 1, tile 1).  Group G - 1 is first read by the B-fragment ring's look-ahead, RD steps before
        // pass 2 (G - 1), i.e. late in pass 2 (G - 1) - 1: one more barrier at the START of that pass orders the carried writes
        // (made three passes earlier); groups 0 .. G - 2 were complete at the layer-boundary barrier as before.
        auto hidden = [&](auto lC) {
            constexpr int l = decltype(lC)::value;            // 1 .. NH
            constexpr int par = l & 1;
            constexpr int NP = 2 * G;                          // tile passes: (group g, tile j) = (i >> 1, i & 1)
            constexpr bool CARRY_IN = true, CARRY_OUT = l < NH;      // layer 1 takes layer 0's last pass
            bf16x8 ring[RD];
#pragma unroll
            for (int d = 0; d < RD; ++d) ring[d] = rd_frag(0, par, d);
            f32x16 accPrev;
            if constexpr (CARRY_IN) accPrev = carry;
            bf16x8 Ho[2];                                      // packed outputs of the previous pass
            bf16x8 w3[4];                                      // layer 3: the LDS-held A fragments of tile 1, k steps 12..15
            f32x16 acc = bias_tile(l, 0);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int j = i & 1;
                f32x16 accNext;
                if constexpr (l == 1) WS_STAMP(8 + i);
                static_assert(RD <= KS, "the ring's look-ahead reaches group G - 1 no earlier than pass 2 (G - 1) - 1");
                if (CARRY_IN && i == 2 * (G - 1) - 1) WS_SYNC();   // the previous layer's carried unit is in LDS on every wave
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int q = i * KS + ks;                 // position in the phase's B-fragment stream
                    if constexpr (l <= 2) mfma_a(acc, Wh[l - 1][j][ks], ring[q % RD]);             // layers 1, 2: the 256 AGPRs
                    else if (j == 1 && ks >= KS - 4) mfma_v(acc, w3[ks - (KS - 4)], ring[q % RD]);  // layer 3's four LDS-held fragments
                    else mfma_v(acc, Wh[l - 1][j][ks], ring[q % RD]);                               // layer 3: arch VGPRs
                    if (q + RD < NP * KS) ring[q % RD] = rd_frag((q + RD) / KS >> 1, par, (q + RD) % KS);
                    if constexpr (l == NH) { if (j == 1 && ks >= KS - 7 && ks < KS - 3) w3[ks - (KS - 7)] = __builtin_bit_cast(bf16x8, bWt[kW0Q + kWoQ + (ks - (KS - 7)) * 64]); }
                    if (i > 0 || CARRY_IN) {
                        // the previous pass's accumulator (pass 0: the previous LAYER's last one), ONE value per step (two in steps 2
                        // and 3): a v_sin next to an MFMA costs ~16 issue cycles, two of them overflow the 24 an MFMA leaves
                        // (tools/micro/mfma_pace.hip).  The compiler does not know the asm statements are MFMAs, so the XDL-write ->
                        // VALU-read wait is kept by hand: two MFMA issues, two LDS issues and 4 idle states.
                        // Where it goes: this layer's output buffer (par ^ 1) — the carried unit into the buffer this layer READS (par).
                        const int pg = i > 0 ? (i - 1) >> 1 : G - 1, pj = i > 0 ? (i - 1) & 1 : 1, wpar = i > 0 ? (par ^ 1) : par;
                        if (ks == 1) asm volatile("s_nop 3" : "+v"(accPrev));
                        if (ks == 2) { act_one(accPrev, Ho, 0); act_one(accPrev, Ho, 1); }
                        if (ks == 3) { act_one(accPrev, Ho, 2); act_one(accPrev, Ho, 3); }
                        if (ks >= 4) act_one(accPrev, Ho, ks);
                        if (ks == 8) wr_frag(pg, wpar, pj, 0, Ho[0]);
                        if (ks == 15) wr_frag(pg, wpar, pj, 1, Ho[1]);
                    }
                    if (ks == 11 && i + 1 < NP) accNext = bias_tile(l, (i + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                accPrev = acc;
                if (i + 1 < NP) acc = accNext;
            }
            if constexpr (l == 1) WS_STAMP(8 + NP);
            if constexpr (CARRY_OUT) {
                carry = accPrev;                               // activated under the next layer's pass 0
            } else {
                // the last hidden layer's last pass: the head reads it in the next phase
                mfma_drain(accPrev);
#pragma unroll
                for (int v = 0; v < 16; ++v) sin_one(accPrev, v);
                cvt_all(accPrev, Ho);
                wr_frag(G - 1, par ^ 1, 1, 0, Ho[0]);
                wr_frag(G - 1, par ^ 1, 1, 1, Ho[1]);
            }
        };
        hidden(IC<1>{});
        WS_STAMP(3);
        WS_SYNC();
        hidden(IC<2>{});
        WS_STAMP(4);
        WS_SYNC();
        WS_STAMP(5);
        if (nextRound < nRounds && w < G) load_raw((nextRound * G + w) * 32 + r);      // in flight under the last hidden phase
        hidden(IC<3>{});
        WS_STAMP(6);
        if (nextRound < nRounds && w < G) stage_raw((int)w);                          // ldsIn was last read before two barriers
        WS_SYNC();
        WS_STAMP(7);
        prevRound = round;
    }
    // the last round's head
    if (prevRound >= 0 && w < G) head(prevRound);
}

// =====================================================================================================
// Near-tie refinement: the marked points (kFlagBit in the stored class) once more, with split-bf16 operands in
// every layer — W = W_hi + W_lo, H = H_hi + H_lo (bf16 each), three MFMAs per k step (lo.hi + hi.lo + hi.hi,
// fp32 accumulate): ~16 mantissa bits per operand instead of 8, so the class agrees with the fp32 reference except
// on ties three orders of magnitude closer (measured: logits within 2e-5 of the range, profiles/r03_inr_refine.txt).
// 2-3 % of the points of a random 4-class head take this path (none of a confident one), at three times the MFMA
// work each.
//
// Dataflow: the streaming kernel's (transposed layers, the accumulator tile is the next layer's B operand; both
// halves of the split stay in registers), simplified where 3 % of the work allows: 4 waves x 32 points per batch,
// ONE wave per SIMD (the hi + lo activations of two layers are 256 registers).  Weights come from the refine image
// one out tile (<= 32 KiB; layer 0: as many tiles as fit) at a time by LDS-DMA into a ring of four LDS buffers,
// THREE tiles ahead of the MFMAs: a tile's MFMAs take 0.6 us and an L2 round trip more than that, so one tile in
// flight left the kernel latency-bound (measured: 2.8 us per tile).  The DMAs span the workgroup barriers: counted
// s_waitcnt vmcnt(2 tiles' pieces) + a raw s_barrier per tile (guide: "Pipelining across barriers"); every tile is
// exactly PERW DMA instructions per wave (a full ring slot: a short tile drags the fragments behind it into slots nobody
// reads) so that the count is one immediate, and in the hidden layers they are issued one at a time between the k steps
// (issue_part); the stream is cyclic (after the head comes layer 0 again), so a batch starts with its first three tiles
// already in LDS.
// Work list: a workgroup scans interleaved 2048-point segments of the class array for marks (16-byte loads), collects the point
// ids in LDS and runs a batch whenever 128 are waiting (the remainder at the end): no global compaction pass, nothing for
// the host to size; the only global atomic is the optional segment ticket (InrArgs::segTicket: the C5 passes).
// =====================================================================================================
constexpr int kRefWaves = 4;
// Diagnostic build (-DMRIRT_REF_STAMPS): s_memtime at the phase boundaries of every workgroup's second batch goes to the logits
// buffer (which then holds no logits): [block][wave][64] uint64, decoded by tools/refine_stamps.py.  Never in the product.
#ifdef MRIRT_REF_STAMPS
#define REF_STAMP(k) do { if (batchNo == 1u && lane == 0) reinterpret_cast<unsigned long long*>(a.logits)[((size_t)blockIdx.x * 4 + waveS) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define REF_STAMP(k) do { } while (0)
#endif

template <int HID, int KT0, bool SIREN, bool AUG>
__global__ __launch_bounds__(256, 1) void inr_refine_kernel(const InrArgs a) {
    constexpr int KT = HID / 32;
    constexpr bool L0SPLIT = !AUG;                       // AUG: layer 0's split is folded into its k axis already
    constexpr int NK0 = KT0 * 2, NKH = KT * 2;           // k steps of an out tile: layer 0 / the wider layers
    constexpr int F0 = NK0 * (L0SPLIT ? 2 : 1), FH = NKH * 2;   // fragments of an out tile
    constexpr int FMAX = F0 > FH ? F0 : FH;
    constexpr int C0 = (FMAX / F0) < KT ? (FMAX / F0 > 0 ? FMAX / F0 : 1) : KT;   // layer-0 out tiles per DMA tile
    static_assert(KT % C0 == 0 && C0 * F0 <= FMAX && FMAX % kRefWaves == 0, "layer-0 chunking");
    constexpr int PERW = FMAX / kRefWaves;               // DMA instructions per wave per tile (always this many)
    constexpr int NBUF = 4, AHEAD = NBUF - 1;            // LDS ring; tiles in flight ahead of the one being read
    constexpr int RD = 4;                                // k steps of A fragments in registers ahead of their MFMAs
    constexpr int kScanV = 8, kSeg = 256 * kScanV;       // a scan step reads 2048 classes: eight consecutive ones (16 bytes) per thread
    constexpr int kBatch = kRefWaves * 32;
    constexpr int kListCap = kBatch + kSeg;
    constexpr int kBiasQ = kMaxLayers * 256 / 4, kTabQ = 128, kRawQ = kRefWaves * 32 * kRawStride / 4;
    constexpr int kWQ = NBUF * FMAX * 64;
    __shared__ uint4 ldsAll[kWQ + kBiasQ + kTabQ + kRawQ + kListCap / 4 + 1];       // ONE LDS object (guide: a second one de-pipelines the DMA); the last uint4: listCount[0..1]
    const float4* ldsBias = reinterpret_cast<const float4*>(ldsAll + kWQ);
    float4* ldsTab = reinterpret_cast<float4*>(ldsAll + kWQ + kBiasQ);
    float* ldsRaw = reinterpret_cast<float*>(ldsAll + kWQ + kBiasQ + kTabQ);
    uint32_t* list = reinterpret_cast<uint32_t*>(ldsAll + kWQ + kBiasQ + kTabQ + kRawQ);
    uint32_t* listCount = list + kListCap;

    int64_t nPts = a.n;
    if (a.nDev != nullptr) { const int64_t nd = (int64_t)*a.nDev; nPts = nd < nPts ? nd : nPts; }
    if (nPts <= 0 || (int64_t)blockIdx.x * kSeg >= nPts) return;       // uniform

    // feature table and biases: as inr_forward_kernel
    if (a.kind < 2 && threadIdx.x < 128) {
        const uint32_t f = threadIdx.x;
        uint32_t src = kRawStride - 1, trig = 0;
        float mult = 0.0f, phase = 0.0f;
        if (f < 3) src = f;
        else if (f < a.L.inDim) {
            uint32_t g = f - 3;
            if (a.kind == MRIRT_INR_FOURIER_RELU && g < 6 * a.K) {
                const uint32_t axis = g / (2 * a.K), rem = g % (2 * a.K);
                const bool isSin = rem < a.K;
                src = axis; trig = 1;
                mult = (float)((isSin ? rem : rem - a.K) + 1) * 0.5f;
                phase = isSin ? 0.0f : 0.25f;
            } else {
                if (a.kind == MRIRT_INR_FOURIER_RELU) g -= 6 * a.K;
                src = 3 + g;
            }
        }
        if constexpr (AUG) {
            const uint32_t kk = f & 15u, in = a.L.inDim;
            src = kRawStride - 1; trig = 0;
            if (f < 16) { if (kk < in) src = kk; else if (kk < 2 * in) { src = kk - in; trig = 2; } }
            else if (f < 32 && kk < in) src = kk;
        }
        ldsTab[f] = make_float4(mult, phase, __builtin_bit_cast(float, src), __builtin_bit_cast(float, trig));
    }
    {
        const uint32_t nq = (a.L.biasOff[a.L.numLayers - 1] + 32) / 4;
        for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) {
            uint32_t l = 0;
            while (l + 1 < a.L.numLayers && 4 * i >= a.L.biasOff[l + 1]) ++l;
            float4 b = reinterpret_cast<const float4*>(a.bias)[i];
            const float sc = a.L.bscale[l];
            b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
            ldsAll[kWQ + i] = __builtin_bit_cast(uint4, b);
        }
    }
    if (threadIdx.x == 0) *listCount = 0u;

    const uint32_t lane = threadIdx.x & 63u, r = lane & 31u, h = lane >> 5;
    const uint32_t waveS = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* __restrict__ wr = a.wpack + (size_t)ref_base_frag_dev(a.L) * 64;      // the refine image

    // ---- the cyclic tile stream: layer 0 in chunks of C0 out tiles, then one out tile at a time, the head, and again ----
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)ldsAll;
    uint32_t isLayer = 0, isTile = 0, isFrag = a.L.refFragOff[0];     // the next tile to ISSUE (scalars)
    uint32_t ringIssue = 0, ringRead = 0;                              // ring slots: next to fill / being read
    bool primed = false;
    uint32_t batchNo = 0;
    // One tile's DMAs in PERW parts, so that the hidden layers can place them in the shadows of their MFMAs: a lone wave issues
    // an instruction every ~8 cycles, and the all-at-once form with its per-DMA address arithmetic (64 instructions) held the
    // MFMAs up for 520 cycles per tile (s_memtime stamps, round 4).  Wave w moves fragments w * PERW .. w * PERW + PERW - 1 of the
    // tile into the slots of the same numbers: ONE address register pair and one M0 value per four parts plus the instruction's
    // immediate offset, which the hardware adds to the global AND the LDS address.  Every tile moves a full ring slot (FMAX
    // fragments): a short tile (layer 0's) drags the fragments that follow it into slots nobody reads, and the image ends in
    // kRefSlackFrags of slack for the last one.
    static_assert(FMAX <= kRefSlackFrags, "the slack after the refine image covers one ring slot");
    const char* curSrc = nullptr;                                     // this wave's first fragment of the tile being issued (+ lane * 16)
    uint32_t curDst = 0;                                              // its ring slot (LDS byte address)
    auto issue_begin = [&]() {
        curSrc = reinterpret_cast<const char*>(wr) + ((size_t)(isFrag + waveS * PERW) << 10) + (lane << 4);
        curDst = __builtin_amdgcn_readfirstlane(ldsBase + ((ringIssue * FMAX + waveS * PERW) << 10));
    };
    auto issue_part = [&](int i) {                                    // (a constant once the caller's loop is unrolled)
        // Issued as inline asm ON PURPOSE: told about an LDS-DMA, hipcc makes the next ds_read whose address it cannot
        // tell apart from the DMA's target (a ring slot chosen at run time) wait vmcnt(0) — the whole look-ahead drained
        // once per tile (seen in the ISA of the builtin form).  The waits for these loads are the counted ones in
        // tile_done(); hipcc's own counted waits (for the batch's input loads) can only over-wait, never under-wait.
        const char* src = curSrc + (i >> 2) * 4096;
        const uint32_t dst = curDst + (uint32_t)((i >> 2) * 4096);
        switch (i & 3) {
            case 0:  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0"); break;
            case 1:  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024" :: "v"(src), "s"(dst) : "memory", "m0"); break;
            case 2:  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:2048" :: "v"(src), "s"(dst) : "memory", "m0"); break;
            default: asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:3072" :: "v"(src), "s"(dst) : "memory", "m0"); break;
        }
    };
    auto issue_end = [&]() {
        const uint32_t last = a.L.numLayers - 1;
        isFrag += isLayer == 0 ? (uint32_t)(C0 * F0) : (uint32_t)FH;
        const uint32_t tilesHere = isLayer == 0 ? (uint32_t)(KT / C0) : (isLayer == last ? 1u : (uint32_t)KT);
        if (++isTile == tilesHere) { isTile = 0; if (++isLayer > last) { isLayer = 0; isFrag = a.L.refFragOff[0]; } }
        ringIssue = ringIssue + 1 == NBUF ? 0u : ringIssue + 1;
    };
    auto issue_tile = [&]() {
        issue_begin();
#pragma unroll
        for (int i = 0; i < PERW; ++i) issue_part(i);
        issue_end();
    };
    // end of a tile: this wave's pieces of the NEXT tile have landed (the two after it may still be in flight), its own
    // LDS reads of this tile have retired; after the barrier that holds for every wave
    auto tile_done = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PERW * (AHEAD - 1)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ringRead = ringRead + 1 == NBUF ? 0u : ringRead + 1;
    };
    auto frag_at = [&](int f) { return __builtin_bit_cast(bf16x8, ldsAll[(ringRead * FMAX + f) * 64 + lane]); };

    // ---- one batch: list[off .. off + m) ---------------------------------------------------------------------
    auto run_batch = [&](uint32_t off, uint32_t m) {
        const uint32_t slot = waveS * 32 + r;
        const bool valid = slot < m;
        const int64_t p = (int64_t)list[off + (valid ? slot : 0u)];

        // layer-0 B operands (the streaming kernel's load_inputs, for an arbitrary point id)
        bf16x8 xin_hi[KT0][2], xin_lo[KT0][2];
        if (a.kind < 2) {
            float* raw = ldsRaw + waveS * 32 * kRawStride + r;
            if (h == 0) {
                float c[3], mm[kMaxMods];
                const uint32_t M = a.M;
                if (a.volume) {
                    const uint32_t k = (uint32_t)(p % a.D), j = (uint32_t)((p / a.D) % a.W), i = (uint32_t)(p / ((int64_t)a.D * a.W));
                    c[0] = (float)(((double)i / (double)(a.H - 1)) * 2.0 - 1.0);
                    c[1] = (float)(((double)j / (double)(a.W - 1)) * 2.0 - 1.0);
                    c[2] = (float)(((double)k / (double)(a.D - 1)) * 2.0 - 1.0);
                    const size_t hwd = (size_t)a.H * a.W * a.D;
#pragma unroll
                    for (uint32_t g = 0; g < kMaxMods; ++g) mm[g] = M ? a.feats[(size_t)(g < M ? g : M - 1) * hwd + p] : 0.0f;
                } else {
                    c[0] = a.coords[p * 3 + 0]; c[1] = a.coords[p * 3 + 1]; c[2] = a.coords[p * 3 + 2];
#pragma unroll
                    for (uint32_t g = 0; g < kMaxMods; ++g) mm[g] = M ? a.feats[p * (int64_t)M + (g < M ? g : M - 1)] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) raw[32 * k] = c[k];
#pragma unroll
                for (int k = 0; k < kMaxMods; ++k) raw[32 * (3 + k)] = mm[k];
                raw[32 * (kRawStride - 1)] = 0.0f;
            }
#pragma unroll
            for (int t = 0; t < KT0; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 d = ldsTab[32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
                        const float x = raw[32 * __builtin_bit_cast(uint32_t, d.z)];
                        const uint32_t mode = __builtin_bit_cast(uint32_t, d.w);
                        const float v = mode == 1u ? __builtin_amdgcn_sinf(__builtin_fmaf(x, d.x, d.y)) : x;
                        const __bf16 hi = (__bf16)v;
                        const __bf16 lo = (__bf16)(v - (float)hi);
                        xin_hi[t][s][j] = (AUG && mode == 2u) ? lo : hi;
                        xin_lo[t][s][j] = lo;
                    }
                }
        } else {
#pragma unroll
            for (int t = 0; t < KT0; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t f = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                        float v = a.feats[p * (int64_t)a.L.inDim + (f < a.L.inDim ? f : 0u)];
                        v = f < a.L.inDim ? v : 0.0f;
                        const __bf16 hi = (__bf16)v;
                        xin_hi[t][s][j] = hi;
                        xin_lo[t][s][j] = (__bf16)(v - (float)hi);
                    }
        }
        // the inputs are in registers (the compiler has waited for every ordinary load above, and with it for any DMA
        // still in flight); from here to the end of the batch the only vector-memory operations in flight are DMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!primed) {                                   // first batch of this workgroup: start the stream
#pragma unroll
            for (int d = 0; d < AHEAD; ++d) issue_tile();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PERW * (AHEAD - 1)) : "memory");
            primed = true;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // first tile in LDS for every wave (and table + biases + list on the first batch)

        // the activations of two layers, hi and lo halves of the split, as PACKED 32-bit words (two bf16 each; word j of
        // [tile][0..7] = elements 2j, 2j + 1 of the tile's 16 per lane).  Written a word at a time by act_pair: element-wise
        // inserts into bf16x8 made hipcc keep every value as an fp32 register until its partner arrived and convert again
        // at the layer's end (round 4: 417 + 161 registers, 128 v_cvt_pk and as many v_accvgpr moves per layer).
        uint32_t Hc_hi[KT][8], Hc_lo[KT][8], Hn_hi[KT][8], Hn_lo[KT][8];
        auto b_frag = [&](const uint32_t (&hw)[8], int s2) {
            return __builtin_bit_cast(bf16x8, make_uint4(hw[4 * s2], hw[4 * s2 + 1], hw[4 * s2 + 2], hw[4 * s2 + 3]));
        };
        auto bias_tile = [&](uint32_t layerOff, int o) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = ldsBias[(layerOff + 32 * o + 8 * g + 4 * h) >> 2];
                acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
            }
            return acc;
        };
        // activation of elements i, i + 1 (i even), then the split of the results: hi = bf16(x), lo = bf16(x - hi)
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        auto act_pair = [&](const f32x16& acc, int o, int i) {
            float x0 = acc[i], x1 = acc[i + 1];
            if constexpr (SIREN) { x0 = __builtin_amdgcn_sinf(x0); x1 = __builtin_amdgcn_sinf(x1); }
            else { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
            const bf16x2 hi = { (__bf16)x0, (__bf16)x1 };
            const uint32_t hp = __builtin_bit_cast(uint32_t, hi);
            const bf16x2 lo = { (__bf16)(x0 - __builtin_bit_cast(float, hp << 16)), (__bf16)(x1 - __builtin_bit_cast(float, hp & 0xffff0000u)) };
            Hn_hi[o][i >> 1] = hp;
            Hn_lo[o][i >> 1] = __builtin_bit_cast(uint32_t, lo);
        };
        auto activate = [&](const f32x16& acc, int o) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) act_pair(acc, o, i);
        };
        // the three products of one k step (small terms first)
        auto mfma3 = [&](f32x16& acc, const bf16x8& whi, const bf16x8& wlo, const bf16x8& bhi, const bf16x8& blo) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, bhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, blo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, bhi, acc, 0, 0, 0);
        };

        // ---- layer 0 -----------------------------------------------------------------------------------------
        REF_STAMP(0);
#ifdef MRIRT_REF_STAMPS
        if (batchNo == 2u && lane == 0) reinterpret_cast<unsigned long long*>(a.logits)[((size_t)blockIdx.x * 4 + waveS) * 64 + 6] = __builtin_amdgcn_s_memtime();
#endif
        {
            const uint32_t b0 = a.L.biasOff[0];
#pragma unroll
            for (int c = 0; c < KT / C0; ++c) {
                issue_tile();
#pragma unroll
                for (int oo = 0; oo < C0; ++oo) {
                    const int o = c * C0 + oo;
                    f32x16 acc = bias_tile(b0, o);
#pragma unroll
                    for (int q = 0; q < NK0; ++q) {
                        const int t = q >> 1, s2 = q & 1;
                        if constexpr (L0SPLIT) mfma3(acc, frag_at(oo * F0 + 2 * q), frag_at(oo * F0 + 2 * q + 1), xin_hi[t][s2], xin_lo[t][s2]);
                        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_at(oo * F0 + q), xin_hi[t][s2], acc, 0, 0, 0);
                    }
                    activate(acc, o);
                }
                tile_done();
            }
        }
        // ---- hidden layers -----------------------------------------------------------------------------------
        REF_STAMP(1);
        for (uint32_t l = 1; l + 1 < a.L.numLayers; ++l) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) { Hc_hi[t][j] = Hn_hi[t][j]; Hc_lo[t][j] = Hn_lo[t][j]; }
            const uint32_t bl = a.L.biasOff[l];
            f32x16 accPrev;                              // finished tile o - 1: activated and split under tile o's MFMAs
            f32x16 biasNext;                             // tile o + 1's biases (its initial accumulator), read under tile o's MFMAs
#pragma unroll
            for (int o = 0; o < KT; ++o) {
                if (l == 2) REF_STAMP(8 + 3 * o);
                issue_begin();
                if (l == 2 && o == 2) REF_STAMP(40);
                f32x16 acc = o == 0 ? bias_tile(bl, 0) : biasNext;
                bf16x8 rhi[RD], rlo[RD];
#pragma unroll
                for (int d = 0; d < RD; ++d) if (d < NKH) { rhi[d] = frag_at(2 * d); rlo[d] = frag_at(2 * d + 1); }
#pragma unroll
                for (int q = 0; q < NKH; ++q) {
                    mfma3(acc, rhi[q % RD], rlo[q % RD], b_frag(Hc_hi[q >> 1], q & 1), b_frag(Hc_lo[q >> 1], q & 1));
                    if (l == 2 && o == 2) REF_STAMP(41 + q);
                    if (q + RD < NKH) { rhi[q % RD] = frag_at(2 * (q + RD)); rlo[q % RD] = frag_at(2 * (q + RD) + 1); }
#pragma unroll
                    for (int i = q * PERW / NKH; i < (q + 1) * PERW / NKH; ++i) issue_part(i);      // the tile three ahead
                    if (q == NKH / 2 && o + 1 < KT) biasNext = bias_tile(bl, o + 1);
                    if (o > 0) {                     // the pairs complete by the end of this k step
#pragma unroll
                        for (int pr = (q * 16 / NKH) / 2; pr < ((q + 1) * 16 / NKH) / 2; ++pr) act_pair(accPrev, o - 1, 2 * pr);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the slices where they are written: one per k step
                }
                issue_end();
                accPrev = acc;
                if (l == 2) REF_STAMP(9 + 3 * o);
                tile_done();
                if (l == 2) REF_STAMP(10 + 3 * o);
            }
            activate(accPrev, KT - 1);                   // the layer's last tile
            REF_STAMP(1 + l);
        }
        // ---- head ----------------------------------------------------------------------------------------------
        {
            REF_STAMP(32);
            issue_tile();
            REF_STAMP(33);
            f32x16 acc = bias_tile(a.L.biasOff[a.L.numLayers - 1], 0);
            bf16x8 rhi[RD], rlo[RD];
#pragma unroll
            for (int d = 0; d < RD; ++d) if (d < NKH) { rhi[d] = frag_at(2 * d); rlo[d] = frag_at(2 * d + 1); }
#pragma unroll
            for (int q = 0; q < NKH; ++q) {
                mfma3(acc, rhi[q % RD], rlo[q % RD], b_frag(Hn_hi[q >> 1], q & 1), b_frag(Hn_lo[q >> 1], q & 1));
                if (q + RD < NKH) { rhi[q % RD] = frag_at(2 * (q + RD)); rlo[q % RD] = frag_at(2 * (q + RD) + 1); }
            }
            REF_STAMP(34);
            float best = -INFINITY, second = -INFINITY;
            uint32_t bestc = 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t cls = (i & 3) + 8 * (i >> 2) + 4 * h;
                top2_push(cls < a.L.outDim ? acc[i] : -INFINITY, cls, best, second, bestc);
            }
            const float ob = __shfl_xor(best, 32);
            const uint32_t oc = __shfl_xor(bestc, 32);
            if (ob > best || (ob == best && oc < bestc)) { best = ob; bestc = oc; }
            REF_STAMP(35);
            tile_done();                                 // (before the stores: they would count against the DMA budget)
            REF_STAMP(5);
            ++batchNo;
#ifdef MRIRT_REF_STAMPS
            if (false) {
#else
            if (a.logits != nullptr && valid) {
#endif
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t cls = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (cls < a.L.outDim) a.logits[p * a.L.outDim + cls] = acc[i];
                }
            }
            if (a.argmax != nullptr && valid && h == 0) a.argmax[p] = (int16_t)bestc;       // the clean class
        }
    };

    // ---- scan for marks, batch as they accumulate ------------------------------------------------------------
    uint32_t cnt = 0;                                    // replicated in every thread: entries waiting in `list`
    __syncthreads();
    const bool vecOK = (reinterpret_cast<uintptr_t>(a.argmax) & 15u) == 0;       // (a sliced tensor: element loads)
    for (int64_t seg = blockIdx.x; seg * kSeg < nPts; ) {
        const int64_t i0 = seg * kSeg + (int64_t)threadIdx.x * kScanV;
        // Eight classes per thread in ONE 16-byte load (round 4: four strided 2-byte loads per thread made the scan of 67 M classes
        // 1.2 ms — half of what the whole second pass cost; now 0.1 ms)
        uint32_t w[4] = { 0u, 0u, 0u, 0u };
        if (!a.refineAll && i0 < nPts) {
            if (vecOK && i0 + kScanV <= nPts) {
                const uint4 q = *reinterpret_cast<const uint4*>(a.argmax + i0);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < kScanV; ++j)
                    if (i0 + j < nPts) w[j >> 1] |= (uint32_t)(uint16_t)a.argmax[i0 + j] << (16 * (j & 1));
            }
        }
        // the segment after this one: the next of the round-robin deal, or the next nobody has taken yet
        if (a.segTicket != nullptr && threadIdx.x == 0) listCount[1] = atomicAdd(a.segTicket, 1u);
        uint64_t mk[kScanV];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < kScanV; ++j) {
            const bool marked = i0 + j < nPts && (a.refineAll != 0u || ((w[j >> 1] >> (16 * (j & 1))) & (uint32_t)kFlagBit) != 0u);
            mk[j] = __ballot(marked);
            total += (uint32_t)__popcll(mk[j]);
        }
        if (total != 0) {                                // wave-uniform: one LDS atomic per wave and step
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(listCount, total);
            wbase = __shfl(wbase, 0);
#pragma unroll
            for (int j = 0; j < kScanV; ++j) {
                if ((mk[j] >> lane) & 1ull)
                    list[wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[j], 0u))] = (uint32_t)(i0 + j);
                wbase += (uint32_t)__popcll(mk[j]);
            }
        }
        __syncthreads();
        cnt = *listCount;
        const int64_t nextSeg = a.segTicket != nullptr ? (int64_t)gridDim.x + (int64_t)listCount[1] : seg + (int64_t)gridDim.x;
        const bool lastSeg = nextSeg * kSeg >= nPts;                 // after the last segment the remainder goes as a short batch
        while (cnt >= (uint32_t)kBatch || (lastSeg && cnt > 0)) {    // (ONE call site: the batch body is inlined once)
            const uint32_t m = cnt < (uint32_t)kBatch ? cnt : (uint32_t)kBatch;
            run_batch(cnt - m, m);
            cnt -= m;
        }
        __syncthreads();                                 // every wave has read its list entries (and the next segment's number)
        if (threadIdx.x == 0) *listCount = cnt;
        __syncthreads();
        seg = nextSeg;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's look-ahead DMAs: nothing may be in flight at exit
}

// nets the weight-stationary kernel takes: the 7-input SIREN with four 256-wide layers and <= 4 classes, points mode
static bool ws_eligible(const InrArgs& a) {
    return a.kind == MRIRT_INR_SIREN && a.L.aug0 && a.L.hidden == 256 && a.L.numLayers == 5 && a.L.outDim <= 4 &&
           a.M == 4 && !a.volume && (reinterpret_cast<uintptr_t>(a.feats) & 15u) == 0;
}

static int launch_inr_ws(const InrArgs& a, hipStream_t s) {
    int dev = 0, cus = 0;
    MRIRT_HIP(hipGetDevice(&dev));
    MRIRT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t rounds = (a.n + kWsGroups * 32 - 1) / (kWsGroups * 32);
    const int64_t nres = cus > 0 ? cus : 256;                  // one workgroup per CU (149 KiB of LDS, 512 registers)
    const dim3 grid((uint32_t)(rounds < nres ? rounds : nres)), block(256);
    hipLaunchKernelGGL(inr_ws_kernel<true>, grid, block, 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

template <int HID>
static int launch_inr_kt0(const InrArgs& a, hipStream_t s) {
    const int64_t groups = (a.n + kInrWaves * 32 - 1) / (kInrWaves * 32);
    const bool siren = a.kind == MRIRT_INR_SIREN || a.kind == 3u;
    void (*kern)(const InrArgs) = a.L.kt0 == 1
        ? (siren ? (a.L.aug0 ? inr_forward_kernel<HID, 1, true, true> : inr_forward_kernel<HID, 1, true, false>)
                 : inr_forward_kernel<HID, 1, false, false>)
        : (siren ? inr_forward_kernel<HID, 4, true, false> : inr_forward_kernel<HID, 4, false, false>);
    bool res = false;
    if constexpr (HID <= 64) {
        if (a.L.totalFrags <= (uint32_t)kResidentFrags) {
            res = true;
            kern = a.L.kt0 == 1
                ? (siren ? (a.L.aug0 ? inr_forward_kernel<HID, 1, true, true, true> : inr_forward_kernel<HID, 1, true, false, true>)
                         : inr_forward_kernel<HID, 1, false, false, true>)
                : (siren ? inr_forward_kernel<HID, 4, true, false, true> : inr_forward_kernel<HID, 4, false, false, true>);
        }
    }
    // persistent workgroups: as many as are resident at once (the 4 x 256 nets: one per CU, 150 KB of LDS)
    // cached per device and kernel family; a relaxed atomic: two threads racing here compute the same number
    constexpr int kMaxDev = 16;
    static std::atomic<int> resident[kMaxDev][2][2][3];
    int dev = 0;
    MRIRT_HIP(hipGetDevice(&dev));
    std::atomic<int>& slot = resident[dev >= 0 && dev < kMaxDev ? dev : 0][res ? 1 : 0][a.L.kt0 == 1 ? 0 : 1][siren ? (a.L.aug0 ? 2 : 1) : 0];
    int nres = (dev >= 0 && dev < kMaxDev) ? slot.load(std::memory_order_relaxed) : 0;
    if (nres == 0) {
        int cus = 0, perCu = 0;
        MRIRT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        MRIRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kern, kInrWaves * 64, 0));
        nres = (cus > 0 ? cus : 256) * (perCu > 0 ? perCu : 1);
        if (dev >= 0 && dev < kMaxDev) slot.store(nres, std::memory_order_relaxed);
    }
    const dim3 grid((uint32_t)(groups < nres ? groups : nres)), block(kInrWaves * 64);
    hipLaunchKernelGGL(kern, grid, block, 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

static int launch_inr_main(const InrArgs& a, hipStream_t s) {
    if (ws_eligible(a) && !(a.flags & MRIRT_INR_NO_WEIGHT_STATIONARY) && !debug_env("MRIRT_INR_NO_WS")) return launch_inr_ws(a, s);
    if ((a.n + kInrWaves * 32 - 1) / (kInrWaves * 32) >= (1ll << 31)) return MRIRT_ERR_ARG;
    switch (a.L.hidden) {
        case 32: return launch_inr_kt0<32>(a, s);
        case 64: return launch_inr_kt0<64>(a, s);
        case 128: return launch_inr_kt0<128>(a, s);
        default: return launch_inr_kt0<256>(a, s);
    }
}

template <int HID>
static int launch_refine_hid(const InrArgs& a, hipStream_t s) {
    const bool siren = a.kind == MRIRT_INR_SIREN || a.kind == 3u;
    void (*kern)(const InrArgs) = a.L.kt0 == 1
        ? (siren ? (a.L.aug0 ? inr_refine_kernel<HID, 1, true, true> : inr_refine_kernel<HID, 1, true, false>)
                 : inr_refine_kernel<HID, 1, false, false>)
        : (siren ? inr_refine_kernel<HID, 4, true, false> : inr_refine_kernel<HID, 4, false, false>);
    int dev = 0, cus = 0;
    MRIRT_HIP(hipGetDevice(&dev));
    MRIRT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t segs = (a.n + 2047) / 2048, nres = cus > 0 ? cus : 256;       // one workgroup per CU (512 registers per wave); 2048-class scan steps
    hipLaunchKernelGGL(kern, dim3((uint32_t)(segs < nres ? segs : nres)), dim3(kRefWaves * 64), 0, s, a);
    MRIRT_HIP(hipGetLastError());
    return MRIRT_OK;
}

// the split-bf16 pass over the marked points of a.argmax (or over every point: a.refineAll)
static int launch_refine(const InrArgs& a, hipStream_t s) {
    if (a.n >= (1ll << 32)) return MRIRT_ERR_ARG;        // point ids travel as 32-bit words
    switch (a.L.hidden) {
        case 32: return launch_refine_hid<32>(a, s);
        case 64: return launch_refine_hid<64>(a, s);
        case 128: return launch_refine_hid<128>(a, s);
        default: return launch_refine_hid<256>(a, s);
    }
}

// near-tie marking + refinement can be switched off for A/B measurements (MrirtInrDesc.flags; the classes are then the bf16 pass's)
static int launch_inr(InrArgs a, hipStream_t s) {
    if (a.n <= 0) return MRIRT_OK;
    if (a.refineAll) return launch_refine(a, s);
    const bool off = (a.flags & MRIRT_INR_NO_REFINE) != 0 || debug_env("MRIRT_INR_NO_REFINE");
    const bool mark = a.argmax != nullptr && a.tie != nullptr && !off;
    if (!mark) a.tie = nullptr;
    else {
        float sig = a.tieSigmas > 0.0f ? a.tieSigmas : kTieSigmas;
#ifdef MRIRT_DEBUG_ENV
        if (const char* e = getenv("MRIRT_INR_TIE_SIGMAS")) { const float v = (float)atof(e); if (v > 0.0f) sig = v; }
#endif
        a.tieScale = sig * 1.41421356f;                  // the gap of two logits carries two errors
    }
    int rc = launch_inr_main(a, s);
    if (rc != MRIRT_OK || !mark || (a.flags & MRIRT_INR_MARK_ONLY) != 0) return rc;
    return launch_refine(a, s);
}

static int fill_args(const MrirtInrDesc* d, InrArgs& a) {
    int rc = make_layout(d, a.L);
    if (rc != MRIRT_OK) return rc;
    if (!d->weights || !d->biases) return MRIRT_ERR_NULL;
    a.kind = d->kind; a.K = d->fourierFreqs; a.M = d->numMods; a.w0 = d->w0;
    a.wpack = static_cast<const uint4*>(d->weights);
    a.bias = d->biases;
    a.coords = nullptr; a.feats = nullptr; a.n = 0; a.nDev = nullptr; a.volume = 0; a.H = a.W = a.D = 1;
    a.logits = nullptr; a.argmax = nullptr;
    a.tie = reinterpret_cast<const float*>(a.wpack + (size_t)tail_frag(a.L) * 64);     // written by mrirt_inr_pack_weights
    a.tieScale = 0.0f; a.refineAll = 0; a.segTicket = nullptr;
    a.flags = d->flags; a.tieSigmas = d->tieSigmas;
    return MRIRT_OK;
}

// Internal (mrirt_host.h): the forward pass with the point count read from device memory — used by the
// chunked C5 render (brats_march.hip), whose producer kernel sizes each chunk's batch on the device.
int inr_forward_dev_n(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t nMax,
                      const uint32_t* nDev, int16_t* argmax, uint32_t* segTicket, hipStream_t s) {
    InrArgs a;
    int rc = fill_args(desc, a);
    if (rc != MRIRT_OK) return rc;
    if (desc->kind >= 2 || !coords || !feats || !argmax || !nDev) return MRIRT_ERR_ARG;
    a.coords = coords; a.feats = feats; a.n = nMax; a.nDev = nDev; a.argmax = argmax; a.segTicket = segTicket;
    return launch_inr(a, s);
}

}  // namespace mrirt

using namespace mrirt;

extern "C" int64_t mrirt_inr_pack_bytes(const MrirtInrDesc* desc) {
    InrLayout L;
    if (make_layout(desc, L) != MRIRT_OK) return 0;
    // main image + slack (the last chunk's DMA over-fetch, never used) + refine image + calibration record
    return (int64_t)(tail_frag(L) + 1) * 1024;
}

namespace mrirt {

__device__ __forceinline__ uint32_t cal_hash(uint32_t x) {         // lowbias32
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float cal_uniform(uint32_t i) { return (float)(cal_hash(i) >> 8) * (1.0f / 16777216.0f); }

// calibration inputs: coords ~ U(-1, 1); modalities ~ the sum of four uniforms, centred and scaled to unit variance
// (z-scored intensities, brats_viewer.py:281-287); raw-x kinds: every input ~ U(-1, 1)
__global__ void inr_cal_inputs_kernel(float* __restrict__ coords, float* __restrict__ feats, uint32_t n, uint32_t featW, uint32_t raw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint32_t k = 0; k < 3; ++k) coords[i * 3 + k] = 2.0f * cal_uniform(i * 131u + k) - 1.0f;
    for (uint32_t k = 0; k < featW; ++k) {
        const uint32_t sd = 0x9e3779b9u + (i * 257u + k) * 4u;
        feats[(size_t)i * featW + k] = raw ? 2.0f * cal_uniform(sd) - 1.0f
                                           : (cal_uniform(sd) + cal_uniform(sd + 1) + cal_uniform(sd + 2) + cal_uniform(sd + 3) - 2.0f) * 1.7320508f;
    }
}

// tail[0] = rms over points and classes of (bf16 pass - split-bf16 pass), [1] = max |logit|, [2] = max |difference|, [3] = points
__global__ __launch_bounds__(256) void inr_cal_reduce_kernel(const float* __restrict__ la, const float* __restrict__ lb, uint32_t count,
                                                             float points, float* __restrict__ tail) {
    __shared__ double ssq[256];
    __shared__ float smx[256], sme[256];
    double q = 0.0;
    float mx = 0.0f, me = 0.0f;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        const float d = la[i] - lb[i];
        q += (double)d * (double)d;
        mx = fmaxf(mx, fabsf(lb[i]));
        me = fmaxf(me, fabsf(d));
    }
    ssq[threadIdx.x] = q; smx[threadIdx.x] = mx; sme[threadIdx.x] = me;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            ssq[threadIdx.x] += ssq[threadIdx.x + o];
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
            sme[threadIdx.x] = fmaxf(sme[threadIdx.x], sme[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tail[0] = (float)sqrt(ssq[0] / (double)count);
        tail[1] = smx[0]; tail[2] = sme[0]; tail[3] = points;
    }
}

// Measures, once per packed network, the error scale of its bf16 pass: kCalPoints pseudo-random inputs through the
// forward kernel and through the split-bf16 kernel, rms difference of the logits -> the calibration record.
static int calibrate(const MrirtInrDesc* d, hipStream_t s) {
    InrArgs a;
    int rc = fill_args(d, a);
    if (rc != MRIRT_OK) return rc;
    const bool raw = d->kind >= 2;
    const uint32_t featW = raw ? d->inDim : (d->numMods > 0 ? d->numMods : 1u), n = kCalPoints;
    const size_t fl = (size_t)n * (3 + featW + 2 * d->outDim);
    float* buf = nullptr;
    MRIRT_HIP(hipMalloc(reinterpret_cast<void**>(&buf), fl * sizeof(float)));
    float* coords = buf; float* feats = coords + (size_t)n * 3; float* la = feats + (size_t)n * featW; float* lb = la + (size_t)n * d->outDim;
    hipLaunchKernelGGL(inr_cal_inputs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, coords, feats, n, featW, raw ? 1u : 0u);
    if (raw) a.M = d->inDim;
    a.coords = raw ? nullptr : coords; a.feats = feats; a.n = n;
    a.logits = la; a.tie = nullptr;
    rc = launch_inr_main(a, s);                          // the bf16 pass (logits only)
    if (rc == MRIRT_OK) { a.logits = lb; a.refineAll = 1; rc = launch_refine(a, s); }
    if (rc == MRIRT_OK) {
        float* tail = const_cast<float*>(reinterpret_cast<const float*>(a.wpack + (size_t)tail_frag(a.L) * 64));
        hipLaunchKernelGGL(inr_cal_reduce_kernel, dim3(1), dim3(256), 0, s, la, lb, n * d->outDim, (float)n, tail);
        if (hipGetLastError() != hipSuccess) rc = MRIRT_ERR_LAUNCH;
    }
    const hipError_t e = hipStreamSynchronize(s);        // load time: the scratch is freed before returning
    (void)hipFree(buf);
    if (e != hipSuccess) return hip_fail(e);
    return rc;
}

}  // namespace mrirt

extern "C" int mrirt_inr_pack_weights(const MrirtInrDesc* desc, const float* w_f32, void* packed, void* stream) {
    InrLayout L;
    int rc = make_layout(desc, L);
    if (rc != MRIRT_OK) return rc;
    if (!w_f32 || !packed) return MRIRT_ERR_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t threads = L.totalFrags * 64;
    hipLaunchKernelGGL(inr_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, w_f32, static_cast<uint4*>(packed), L);
    MRIRT_HIP(hipGetLastError());
    uint4* ref = static_cast<uint4*>(packed) + (size_t)ref_base_frag(L) * 64;
    hipLaunchKernelGGL(inr_pack_ref_kernel, dim3((L.refTotalFrags * 64 + 255) / 256), dim3(256), 0, s, w_f32, ref, L);
    MRIRT_HIP(hipGetLastError());
    // the slack after the refine image (DMA'd into ring slots nobody reads) and the calibration record: zero (= nothing is
    // marked) until the biases are known
    MRIRT_HIP(hipMemsetAsync(ref + (size_t)L.refTotalFrags * 64, 0, ((size_t)kRefSlackFrags + 1) * 1024, s));
    if (desc->weights == packed && desc->biases != nullptr) return calibrate(desc, s);
    return MRIRT_OK;
}

extern "C" int mrirt_inr_calibrate(const MrirtInrDesc* desc, void* stream) {
    if (!desc || !desc->weights || !desc->biases) return MRIRT_ERR_NULL;
    return calibrate(desc, static_cast<hipStream_t>(stream));
}

extern "C" int mrirt_inr_forward_refined(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t n,
                                         float* logits, int16_t* argmax, void* stream) {
    InrArgs a;
    int rc = fill_args(desc, a);
    if (rc != MRIRT_OK) return rc;
    if (n < 0) return MRIRT_ERR_ARG;
    const bool needCoords = desc->kind < 2, needFeats = desc->kind >= 2 || desc->numMods > 0;
    if ((needCoords && !coords) || (needFeats && !feats)) return MRIRT_ERR_NULL;
    if (!logits && !argmax) return MRIRT_ERR_NULL;
    if (desc->kind >= 2) a.M = desc->inDim;
    a.coords = desc->kind < 2 ? coords : nullptr;
    a.feats = feats; a.n = n; a.logits = logits; a.argmax = argmax; a.refineAll = 1;
    return launch_inr(a, static_cast<hipStream_t>(stream));
}

extern "C" int mrirt_inr_forward(const MrirtInrDesc* desc, const float* coords, const float* feats, int64_t n,
                                 float* logits, int16_t* argmax, void* stream) {
    InrArgs a;
    int rc = fill_args(desc, a);
    if (rc != MRIRT_OK) return rc;
    if (n < 0) return MRIRT_ERR_ARG;
    const bool needCoords = desc->kind < 2, needFeats = desc->kind >= 2 || desc->numMods > 0;
    if ((needCoords && !coords) || (needFeats && !feats)) return MRIRT_ERR_NULL;
    if (!logits && !argmax) return MRIRT_ERR_NULL;
    if (desc->kind >= 2) a.M = desc->inDim;              // raw-x kinds: feats IS the input matrix [n][inDim]
    a.coords = desc->kind < 2 ? coords : nullptr;
    a.feats = feats; a.n = n; a.logits = logits; a.argmax = argmax;
    return launch_inr(a, static_cast<hipStream_t>(stream));
}

extern "C" int mrirt_inr_predict_volume(const MrirtInrDesc* desc, const float* mods, const uint32_t hwd[3],
                                        int16_t* pred, void* stream) {
    InrArgs a;
    int rc = fill_args(desc, a);
    if (rc != MRIRT_OK) return rc;
    if (!mods || !hwd || !pred) return MRIRT_ERR_NULL;
    if (desc->kind != MRIRT_INR_FOURIER_RELU && desc->kind != MRIRT_INR_SIREN) return MRIRT_ERR_ARG;
    for (int k = 0; k < 3; ++k) if (hwd[k] < 2) return MRIRT_ERR_DIMS;
    a.feats = mods; a.volume = 1; a.H = hwd[0]; a.W = hwd[1]; a.D = hwd[2];
    a.n = (int64_t)hwd[0] * hwd[1] * hwd[2];
    a.argmax = pred;
    return launch_inr(a, static_cast<hipStream_t>(stream));
}
