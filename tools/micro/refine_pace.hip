// Micro-benchmark for inr_refine_kernel's hidden-layer out tile: 16 k steps x three dependent v_mfma_f32_32x32x16_bf16
// (lo.hi + hi.lo + hi.hi) fed by two ds_read_b128 per step from a 4-deep ring, while the PREVIOUS tile's 16 accumulator
// values are activated (v_sin), split into hi = bf16(x), lo = bf16(x - hi) and parked.  One wave per SIMD, one block per CU,
// as the kernel.  Which arrangement of the ~7 VALU instructions per value costs least beside the MFMAs?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/refine_pace.hip -o /tmp/refine_pace && /tmp/refine_pace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// VARIANT 0: no activation.  1: one value per k step, right where the kernel has it (sched_barrier per step).
// 2: all 16 values after the tile's MFMAs (nothing beside the MFMAs).  3: as 1 without the sched_barrier (compiler's order).
// 4: two values every second step.  5: as 1, sin + one convert only (no split).  6: as 2 on the CURRENT tile's accumulator
// (no accPrev copy).  7: as 1 with the parked results / accPrev forced into AGPRs (what the kernel's register budget does).
// 8: four values every fourth step.  9: all 16 values in the tile's first 4 steps.
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void ktile(const uint4* __restrict__ w, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[32 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) lds[i] = (u32x4){ 0x3f803f80u ^ (unsigned)i, 0x3c003c00u, 0x3f003f00u, 0x3e803e80u };
    __syncthreads();
    bf16x8 Hhi[16], Hlo[16];                          // the layer's B operands (activations of the previous layer)
#pragma unroll
    for (int i = 0; i < 16; ++i) { Hhi[i] = __builtin_bit_cast(bf16x8, w[(i & 7) * 64 + lane]); Hlo[i] = __builtin_bit_cast(bf16x8, w[(8 + (i & 7)) * 64 + lane]); }
    typedef __attribute__((address_space(3))) u32x4 lq;
    lq* base = (lq*)lds + lane;
    constexpr int RD = 4;
    f32x16 accPrev = (f32x16)(0.125f);
    unsigned int nhi[8] = { 0 }, nlo[8] = { 0 };       // the parked results: 16 values -> 8 packed hi + 8 packed lo registers
    auto act_one = [&](const f32x16& src, int i) {
        float x = __builtin_amdgcn_sinf(src[i]);
        const __bf16 hi = (__bf16)x;
        if (VARIANT == 5) { nhi[i >> 1] = (nhi[i >> 1] << 16) | __builtin_bit_cast(unsigned short, hi); return; }
        const __bf16 lo = (__bf16)(x - (float)hi);
        nhi[i >> 1] = (i & 1) ? (nhi[i >> 1] & 0xffffu) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) : (unsigned)__builtin_bit_cast(unsigned short, hi);
        nlo[i >> 1] = (i & 1) ? (nlo[i >> 1] & 0xffffu) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16) : (unsigned)__builtin_bit_cast(unsigned short, lo);
    };
    unsigned int sum = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {                  // 64 tiles
        f32x16 acc = (f32x16)(0.0f);
        bf16x8 rhi[RD], rlo[RD];
#pragma unroll
        for (int d = 0; d < RD; ++d) { rhi[d] = __builtin_bit_cast(bf16x8, base[(2 * d) * 64]); rlo[d] = __builtin_bit_cast(bf16x8, base[(2 * d + 1) * 64]); }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rlo[q % RD], Hhi[q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rhi[q % RD], Hlo[q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rhi[q % RD], Hhi[q], acc, 0, 0, 0);
            if (q + RD < 16) { rhi[q % RD] = __builtin_bit_cast(bf16x8, base[(2 * (q + RD)) * 64]); rlo[q % RD] = __builtin_bit_cast(bf16x8, base[(2 * (q + RD) + 1) * 64]); }
            if (VARIANT == 1 || VARIANT == 3 || VARIANT == 5 || VARIANT == 7) act_one(accPrev, q);
            if (VARIANT == 4 && (q & 1)) { act_one(accPrev, q - 1); act_one(accPrev, q); }
            if (VARIANT == 8 && (q & 3) == 3) { for (int i = q - 3; i <= q; ++i) act_one(accPrev, i); }
            if (VARIANT == 9 && q < 4) { for (int i = 4 * q; i < 4 * q + 4; ++i) act_one(accPrev, i); }
            if (VARIANT != 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (VARIANT == 2) { for (int i = 0; i < 16; ++i) act_one(accPrev, i); }
        if (VARIANT == 6) { for (int i = 0; i < 16; ++i) act_one(acc, i); }
        if (VARIANT == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("" : "+a"(nhi[i])); asm volatile("" : "+a"(nlo[i])); }
            asm volatile("" : "+a"(acc));
        }
        accPrev = acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += nhi[i] ^ nlo[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = (float)sum;
    for (int i = 0; i < 16; ++i) s += accPrev[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT> void run(const char* name, uint4* w, unsigned long long* out, float* sink) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ktile<VARIANT>, dim3(256), dim3(256), 0, 0, w, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s median %.0f cycles per 48-MFMA tile (ideal 1536)\n", name, h[512] / 64.0);
}

int main() {
    uint4* w; unsigned long long* out; float* sink;
    hipMalloc(&w, 20 * 64 * 16); hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 256 * 256 * 4);
    std::vector<unsigned> hw(20 * 64 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u >> 20);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<0>("tile: MFMAs + LDS ring, no activation", w, out, sink);
    run<1>("tile: one value per k step (the kernel's order)", w, out, sink);
    run<7>("tile: one value per k step, results + accumulators in AGPRs", w, out, sink);
    run<3>("tile: one value per k step, compiler's own schedule", w, out, sink);
    run<5>("tile: one value per k step, sin + convert only (no split)", w, out, sink);
    run<4>("tile: two values every second step", w, out, sink);
    run<8>("tile: four values every fourth step", w, out, sink);
    run<9>("tile: all 16 values in the first four steps", w, out, sink);
    run<2>("tile: all 16 values after the MFMAs (previous tile's)", w, out, sink);
    run<6>("tile: all 16 values after the MFMAs (this tile's)", w, out, sink);
    return 0;
}
