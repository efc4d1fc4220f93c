// What the socket's power limit leaves of dense bf16 / fp16 MFMA, in WALL time: one wave per SIMD on every CU runs a dependent chain
// of v_mfma_f32_32x32x16 for ~0.3 s per mode, fed (a) from registers only, (b) with one ds_read_b128 per MFMA (the weight-stationary
// kernel's B ring), (c) with one ds_read_b128 per two MFMAs, (d) as (b) plus the activation's v_sin / convert per step.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool F16>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ w, float* sink, int iters) {
    __shared__ u32x4 lds[16 * 64];
    const int lane = threadIdx.x & 63;
    // RANDOM operands (values in (-2, 2) as bf16 / fp16 bit patterns): what a multiplier draws depends on the bits it toggles —
    // with constant operands this loop reaches 0.96 of the 2.5 PFLOP/s peak, which no real network's data will
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        u32x4 q;
        for (int c = 0; c < 4; ++c) { h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; q[c] = (h & 0x83ff83ffu) | 0x3c003c00u; h = h * 3 + 1; }
        lds[i] = q;
    }
    __syncthreads();
    bf16x8 A[16], Br[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = __builtin_bit_cast(bf16x8, w[i * 64 + lane]);
#pragma unroll
    for (int i = 0; i < 4; ++i) Br[i] = __builtin_bit_cast(bf16x8, w[(16 + i) * 64 + lane]);
    f32x16 acc = (f32x16)(0.0f);
    float x0 = 0.3f * lane;
    typedef __attribute__((address_space(3))) u32x4 lq;
    lq* base = (lq*)lds + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(Br[i & 3]));
            else     asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(Br[i & 3]));
            if (MODE == 1 || MODE == 3) Br[i & 3] = __builtin_bit_cast(bf16x8, base[i * 64]);
            if (MODE == 2 && (i & 1)) Br[(i >> 1) & 3] = __builtin_bit_cast(bf16x8, base[i * 64]);
            if (MODE == 3) x0 = __builtin_amdgcn_sinf(x0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, bool F16> void run(const char* name, uint4* w, float* sink) {
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, F16>), dim3(256), dim3(256), 0, 0, w, sink, iters);
    hipDeviceSynchronize();
    float ms = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {            // grow the launch to ~0.3 s so that the clocks settle
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, F16>), dim3(256), dim3(256), 0, 0, w, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < 300.0f) iters = (int)(iters * 300.0f / (ms > 1.0f ? ms : 1.0f));
    }
    const double flop = 256.0 * 4 * (double)iters * 16 * 32768.0;     // CUs x waves x MFMAs x flop per MFMA
    printf("%-70s %8.1f ms  %7.1f TFLOP/s  (%.3f of 2500)\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15);
}

int main() {
    uint4* w; float* sink;
    hipMalloc(&w, 20 * 64 * 16); hipMalloc(&sink, 256 * 256 * 4);
    {
        unsigned* hw = new unsigned[20 * 64 * 4];
        unsigned h = 12345u;
        for (int i = 0; i < 20 * 64 * 4; ++i) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; hw[i] = (h & 0x83ff83ffu) | 0x3c003c00u; }   // sign + mantissa random, exponent of 1.x
        hipMemcpy(w, hw, 20 * 64 * 16, hipMemcpyHostToDevice);
        delete[] hw;
    }
    run<0, false>("bf16 MFMA chain, operands in registers", w, sink);
    run<1, false>("bf16 + one ds_read_b128 per MFMA", w, sink);
    run<2, false>("bf16 + one ds_read_b128 per two MFMAs", w, sink);
    run<3, false>("bf16 + ds_read per MFMA + v_sin per MFMA", w, sink);
    run<0, true>("fp16 MFMA chain, operands in registers", w, sink);
    run<1, true>("fp16 + one ds_read_b128 per MFMA", w, sink);
    return 0;
}
