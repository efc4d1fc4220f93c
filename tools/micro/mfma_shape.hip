// Micro-benchmark: wall-clock FLOP/s of the two bf16 MFMA shapes in the weight-stationary kernel's inner loop (A fragments in
// registers, one B fragment per 32 MFMA-cycles from LDS, random operands), one wave per SIMD, 256 workgroups, long enough
// for the chip to settle its clock.  The guide (MI355X_MICROARCH.md, DVFS give-back 7) reports the 16x16x32 shape ~1.12-1.15x.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape.hip -o build_exp/mfma_shape && build_exp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool ACT>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ w, float* sink, int iters) {
    __shared__ u32x4 lds[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = __builtin_bit_cast(u32x4, w[(32 + (i >> 6)) * 64 + (i & 63)]);
    __syncthreads();
    bf16x8 A[32], Br[4];
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = __builtin_bit_cast(bf16x8, w[i * 64 + lane]);
    typedef __attribute__((address_space(3))) u32x4 lq;
    lq* base = (lq*)lds + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) Br[i] = __builtin_bit_cast(bf16x8, base[i * 64]);
    f32x16 acc = (f32x16)(0.0f);
    f32x4 a4[4] = { (f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f), (f32x4)(0.0f) };
    f32x16 prev = (f32x16)(0.3f);                      // the previous tile's accumulator, activated under this tile's MFMAs
    float keep = 0.0f;
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {          // one 32x32 tile, k = 256: 16 MFMAs of 32 cycles, one B fragment each
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[ks]), "v"(Br[ks % 4]));
                Br[ks % 4] = __builtin_bit_cast(bf16x8, base[((ks + 4) & 15) * 64]);
                if (ACT) {                              // one sine per 32 MFMA cycles, a packed convert every second one
                    prev[ks] = __builtin_amdgcn_sinf(prev[ks]);
                    if (ks & 1) { bf2 pk = { (__bf16)prev[ks - 1], (__bf16)prev[ks] }; keep += __builtin_bit_cast(float, pk); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((it & 63) == 63) acc = acc * 1e-3f;
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {           // the same tile as 2 x 2 sub-tiles, k steps of 32: 32 MFMAs of 16 cycles,
#pragma unroll                                          // two B fragments (point halves) per k step
                for (int hb = 0; hb < 2; ++hb) {
                    const bf16x8 b = Br[(2 * ks + hb) % 4];
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[hb]) : "a"(A[2 * ks]), "v"(b));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[2 + hb]) : "a"(A[2 * ks + 1]), "v"(b));
                    Br[(2 * ks + hb) % 4] = __builtin_bit_cast(bf16x8, base[((2 * ks + hb + 4) & 15) * 64]);
                    if (ACT) {
                        const int v = 2 * ks + hb;
                        prev[v] = __builtin_amdgcn_sinf(prev[v]);
                        if (v & 1) { bf2 pk = { (__bf16)prev[v - 1], (__bf16)prev[v] }; keep += __builtin_bit_cast(float, pk); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((it & 63) == 63) { a4[0] = a4[0] * 1e-3f; a4[1] = a4[1] * 1e-3f; a4[2] = a4[2] * 1e-3f; a4[3] = a4[3] * 1e-3f; }
        }
    }
    float s = keep;
    for (int i = 0; i < 16; ++i) s += acc[i] + prev[i];
    for (int i = 0; i < 4; ++i) s += a4[i].x + a4[i].y + a4[i].z + a4[i].w;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, bool ACT> void run(const char* name, uint4* w, float* sink) {
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, ACT>), dim3(256), dim3(256), 0, 0, w, sink, iters / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, ACT>), dim3(256), dim3(256), 0, 0, w, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * iters * 16.0 * 32768.0 * 4.0 * 256.0;      // per iteration: 16 (32x32x16) MFMA-equivalents per wave
    printf("%-28s %.2f ms  %.1f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
    uint4* w; float* sink;
    hipMalloc(&w, 48 * 64 * 16); hipMalloc(&sink, 256 * 256 * 4);
    std::vector<unsigned short> hw(48 * 64 * 8);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> d(-1.0f, 1.0f);
    for (auto& x : hw) { float f = d(rng) * 0.25f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("32x32x16, random operands", w, sink);
        run<1, false>("16x16x32, random operands", w, sink);
        run<0, true>("32x32x16 + sin/cvt, random", w, sink);
        run<1, true>("16x16x32 + sin/cvt, random", w, sink);
    }
    hipMemset(w, 0, 48 * 64 * 16);
    run<0, false>("32x32x16, zero operands", w, sink);
    run<1, false>("16x16x32, zero operands", w, sink);
    return 0;
}
