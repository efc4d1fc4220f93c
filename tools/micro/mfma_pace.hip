// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 in a dependent chain, by operand register class and with /
// without a ds_read_b128 + s_waitcnt in every gap.  One wave per SIMD (256-thread block), one block per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_pace.hip -o /tmp/mfma_pace && /tmp/mfma_pace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ w, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = (u32x4){ (unsigned)i, 1u, 2u, 3u };
    __syncthreads();
    bf16x8 A[16], B[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = __builtin_bit_cast(bf16x8, w[i * 64 + lane]);
#pragma unroll
    for (int i = 0; i < 4; ++i) B[i] = __builtin_bit_cast(bf16x8, w[(16 + i) * 64 + lane]);
    f32x16 acc = (f32x16)(0.0f);
    bf16x8 Br[8];
    for (int i = 0; i < 8; ++i) Br[i] = B[i & 3];
    float x0 = 0.3f * lane, x1 = 0.7f * lane;
    typedef __attribute__((address_space(3))) u32x4 lq;
    lq* base = (lq*)lds + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(B[i & 3]));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A[i]), "v"(B[i & 3]));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A[i]), "v"(B[i & 3]));
            if (MODE == 3 || MODE == 4) {   // with an LDS read refilling the B ring every step
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(B[i & 3]));
                B[i & 3] = __builtin_bit_cast(bf16x8, base[((i + it) & 15) * 64]);
                if (MODE == 4) { acc[0] = __builtin_amdgcn_sinf(acc[0] * 0.0f + 1.0f * i); }   // (breaks the chain: not used)
            }
            if (MODE == 7 || MODE == 8 || MODE == 9) {   // immediate-offset LDS read (as the kernel's B ring), ring of 4 / 8
                constexpr int RDm = MODE == 8 ? 8 : 4;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(Br[i % RDm]));
                Br[i % RDm] = __builtin_bit_cast(bf16x8, base[i * 64]);
                if (MODE == 9) { x0 = __builtin_amdgcn_sinf(x0); x1 = __builtin_amdgcn_sinf(x1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 10) {   // accumulator in AGPRs + LDS ring into VGPRs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A[i]), "v"(Br[i % 4]));
                Br[i % 4] = __builtin_bit_cast(bf16x8, base[i * 64]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 11) {   // A in VGPRs, B ring loaded into AGPRs, acc in VGPRs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A[i]), "a"(Br[i % 4]));
                Br[i % 4] = __builtin_bit_cast(bf16x8, base[i * 64]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 12) {   // ds_read only every second step (two MFMAs per B fragment)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(Br[(i >> 1) % 4]));
                if (i & 1) Br[(i >> 1) % 4] = __builtin_bit_cast(bf16x8, base[i * 64]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 13) {   // ds_read_b64 x2 instead of b128
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A[i]), "v"(Br[i % 4]));
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(3))) u32x2 lq2;
                lq2* b2 = (lq2*)lds + lane;
                u32x2 lo = b2[i * 128], hi = b2[i * 128 + 64];
                Br[i % 4] = __builtin_bit_cast(bf16x8, (u32x4){ lo.x, lo.y, hi.x, hi.y });
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[i & 3], acc, 0, 0, 0);
            if (MODE == 6) {                 // builtin + LDS ring
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[i & 3], acc, 0, 0, 0);
                B[i & 3] = __builtin_bit_cast(bf16x8, base[((i + it) & 15) * 64]);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = x0 + x1;
    for (int i = 0; i < 16; ++i) s += acc[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// sin(2 pi x), x in revolutions (what v_sin_f32 takes), as a range-reduced odd minimax polynomial: r = x - rint(x) in
// [-0.5, 0.5], s = r P(r^2), P of degree 3 (odd degree 7): max error 2.5e-4 < 2^-10, below the bf16 rounding that follows
// (VERDICT r3 #3 asked whether this beats v_sin_f32's 8 issue cycles beside the MFMAs).  7 plain VALU instructions per value.
__device__ __forceinline__ float sin_poly(float x) {
    const float r = x - __builtin_rintf(x), r2 = r * r;
    float p = __builtin_fmaf(-56.086395f, r2, 77.93035f);
    p = __builtin_fmaf(p, r2, -41.09373f);
    p = __builtin_fmaf(p, r2, 6.2786355f);
    return r * p;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 sin_poly2(f32x2 x) {       // the same on a register pair: v_pk_add / v_pk_mul / v_pk_fma_f32
    const f32x2 n = { __builtin_rintf(x.x), __builtin_rintf(x.y) };
    const f32x2 r = x - n, r2 = r * r;
    f32x2 p = __builtin_elementwise_fma((f32x2){ -56.086395f, -56.086395f }, r2, (f32x2){ 77.93035f, 77.93035f });
    p = __builtin_elementwise_fma(p, r2, (f32x2){ -41.09373f, -41.09373f });
    p = __builtin_elementwise_fma(p, r2, (f32x2){ 6.2786355f, 6.2786355f });
    return r * p;
}
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 sin_poly_h2(f32x2 x) {        // range reduction in fp32, polynomial in packed f16 (v_pk_fma_f16: one pass)
    const f32x2 n = { __builtin_rintf(x.x), __builtin_rintf(x.y) };
    const f32x2 rf = x - n;
    const h2 r = { (_Float16)rf.x, (_Float16)rf.y }, r2 = r * r;
    h2 p = __builtin_elementwise_fma((h2){ (_Float16)-56.086395f, (_Float16)-56.086395f }, r2, (h2){ (_Float16)77.93035f, (_Float16)77.93035f });
    p = __builtin_elementwise_fma(p, r2, (h2){ (_Float16)-41.09373f, (_Float16)-41.09373f });
    p = __builtin_elementwise_fma(p, r2, (h2){ (_Float16)6.2786355f, (_Float16)6.2786355f });
    return r * p;
}

// The weight-stationary kernel's pass structure: 16 MFMAs into acc while the PREVIOUS pass's accumulator is activated
// (2 v_sin per step in steps 2..9, packed converts), one B fragment per step from LDS.  VARIANT 0: accumulators in
// VGPRs (VALU reads them directly); 1: no activation at all; 2: activation on registers MFMAs never wrote;
// 3: accumulators in AGPRs, read out with v_accvgpr_read.
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void kpass(const uint4* __restrict__ w, unsigned long long* out, float* sink) {
    __shared__ u32x4 lds[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = (u32x4){ (unsigned)i, 1u, 2u, 3u };
    __syncthreads();
    bf16x8 A[16], Br[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = __builtin_bit_cast(bf16x8, w[i * 64 + lane]);
#pragma unroll
    for (int i = 0; i < 4; ++i) Br[i] = __builtin_bit_cast(bf16x8, w[(16 + i) * 64 + lane]);
    f32x16 acc[2] = { (f32x16)(0.0f), (f32x16)(0.0f) }, other = (f32x16)(0.25f);
    typedef __attribute__((address_space(3))) u32x4 lq;
    lq* base = (lq*)lds + lane;
    float keep = 0.0f, keep2 = 0.0f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 bq = (f32x4)(0.0f);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 32; ++it) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {                     // two passes per trip: acc[p] accumulates, acc[p ^ 1] is activated
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (VARIANT == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[p]) : "a"(A[ks]), "v"(Br[ks % 4]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[p]) : "a"(A[ks]), "v"(Br[ks % 4]));
                Br[ks % 4] = __builtin_bit_cast(bf16x8, base[ks * 64]);
                if (VARIANT == 4) {                       // one value per step over all 16 steps: bias add + sin, convert every second step
                    f32x16& src = acc[p ^ 1];
                    if (ks == 0) asm volatile("s_nop 15" : "+v"(src));
                    if ((ks & 3) == 0) bq = __builtin_bit_cast(f32x4, base[(ks >> 2) * 64 + 8]);
                    const float bb = (ks & 3) == 0 ? bq.x : ((ks & 3) == 1 ? bq.y : ((ks & 3) == 2 ? bq.z : bq.w));
                    src[ks] = __builtin_amdgcn_sinf(src[ks] + bb);
                    if (ks & 1) {
                        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                        bf2 pk = { (__bf16)src[ks - 1], (__bf16)src[ks] };
                        keep += __builtin_bit_cast(float, pk);
                    }
                }
                if (VARIANT == 5) {                       // as 4, the sine as 7 plain VALU instructions (sin_poly)
                    f32x16& src = acc[p ^ 1];
                    if (ks == 0) asm volatile("s_nop 15" : "+v"(src));
                    if ((ks & 3) == 0) bq = __builtin_bit_cast(f32x4, base[(ks >> 2) * 64 + 8]);
                    const float bb = (ks & 3) == 0 ? bq.x : ((ks & 3) == 1 ? bq.y : ((ks & 3) == 2 ? bq.z : bq.w));
                    src[ks] = sin_poly(src[ks] + bb);
                    if (ks & 1) {
                        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                        bf2 pk = { (__bf16)src[ks - 1], (__bf16)src[ks] };
                        keep += __builtin_bit_cast(float, pk);
                    }
                }
                if (VARIANT == 6 || VARIANT == 7) {       // two values every second step: packed f32 (6) / packed f16 (7) polynomial
                    f32x16& src = acc[p ^ 1];
                    if (ks == 0) asm volatile("s_nop 15" : "+v"(src));
                    if ((ks & 3) == 0) bq = __builtin_bit_cast(f32x4, base[(ks >> 2) * 64 + 8]);
                    if (ks & 1) {
                        const f32x2 bb = (ks & 2) ? (f32x2){ bq.z, bq.w } : (f32x2){ bq.x, bq.y };
                        const f32x2 xin = (f32x2){ src[ks - 1], src[ks] } + bb;
                        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                        if (VARIANT == 6) {
                            const f32x2 y = sin_poly2(xin);
                            src[ks - 1] = y.x; src[ks] = y.y;
                            bf2 pk = { (__bf16)y.x, (__bf16)y.y };
                            keep += __builtin_bit_cast(float, pk);
                        } else {
                            const h2 y = sin_poly_h2(xin);
                            src[ks - 1] = (float)y.x;
                            bf2 pk = { (__bf16)(float)y.x, (__bf16)(float)y.y };
                            keep += __builtin_bit_cast(float, pk);
                        }
                    }
                }
                if (VARIANT == 8 || VARIANT == 9 || VARIANT == 10) {
                    // 8: the weight-stationary kernel's own order — one v_sin per step and, on odd steps, the packed convert of the
                    //    value the sine has JUST produced (a dependent instruction behind a transcendental: with one wave per SIMD
                    //    nothing else can issue while it waits);
                    // 9: software-pipelined — the convert takes the pair whose sines were issued two and three steps earlier;
                    // 10: as 9 with the sine of step ks issued BEFORE the step's LDS read.
                    f32x16& src = acc[p ^ 1];
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    if (ks == 0) asm volatile("s_nop 15" : "+v"(src));
                    src[ks] = __builtin_amdgcn_sinf(src[ks]);
                    if (VARIANT == 8) {
                        if (ks & 1) { bf2 pk = { (__bf16)src[ks - 1], (__bf16)src[ks] }; keep += __builtin_bit_cast(float, pk); }
                    } else {
                        if (ks & 1) { const int a = (ks + 13) & 15, b = (ks + 14) & 15; bf2 pk = { (__bf16)src[a], (__bf16)src[b] }; keep2 += __builtin_bit_cast(float, pk); }
                    }
                }
                if (VARIANT != 1 && VARIANT < 4 && ks >= 2 && ks <= 9) {
                    f32x16& src = VARIANT == 2 ? other : acc[p ^ 1];
                    if (ks == 2) asm volatile("s_nop 15" : "+v"(src));
                    const float a0 = __builtin_amdgcn_sinf(src[2 * ks - 4]), a1 = __builtin_amdgcn_sinf(src[2 * ks - 3]);
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    bf2 pk = { (__bf16)a0, (__bf16)a1 };
                    keep += __builtin_bit_cast(float, pk);
                    src[2 * ks - 4] = a0 * 0.5f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = keep + keep2;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + other[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VARIANT> void runpass(const char* name, uint4* w, unsigned long long* out, float* sink) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kpass<VARIANT>, dim3(256), dim3(256), 0, 0, w, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-46s median %.1f cycles per 16-MFMA pass (ideal 512)\n", name, h[512] / 64.0);
}

template <int MODE> void run(const char* name, uint4* w, unsigned long long* out, float* sink) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-46s median %.2f cycles per MFMA (min %.2f, max %.2f)\n", name, h[512] / 1024.0, h[0] / 1024.0, h[1023] / 1024.0);
}

int main() {
    uint4* w; unsigned long long* out; float* sink;
    hipMalloc(&w, 20 * 64 * 16); hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 256 * 256 * 4);
    std::vector<unsigned> hw(20 * 64 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u >> 20);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<0>("asm A=agpr B=vgpr C/D=vgpr", w, out, sink);
    run<1>("asm A=agpr B=vgpr C/D=agpr", w, out, sink);
    run<2>("asm A=vgpr B=vgpr C/D=vgpr", w, out, sink);
    run<3>("asm A=agpr C/D=vgpr + ds_read_b128 per step", w, out, sink);
    run<7>("asm + ds_read_b128 imm offset, ring 4", w, out, sink);
    run<8>("asm + ds_read_b128 imm offset, ring 8", w, out, sink);
    run<9>("asm + ds_read imm, ring 4 + 2 v_sin per step", w, out, sink);
    run<10>("asm C/D=agpr + ds_read imm ring 4", w, out, sink);
    run<11>("asm A=vgpr, B ring in AGPRs + ds_read", w, out, sink);
    run<12>("asm + ds_read every 2nd MFMA", w, out, sink);
    run<13>("asm + 2 x ds_read_b64 per step", w, out, sink);
    run<5>("builtin", w, out, sink);
    runpass<1>("pass: no activation", w, out, sink);
    runpass<0>("pass: activation of the other VGPR accumulator", w, out, sink);
    runpass<2>("pass: activation of registers MFMA never wrote", w, out, sink);
    runpass<3>("pass: accumulators in AGPRs (accvgpr_read)", w, out, sink);
    runpass<4>("pass: 1 value/step x 16, bias add + sin + cvt", w, out, sink);
    runpass<8>("pass: 1 sin/step, cvt right behind its sine", w, out, sink);
    runpass<9>("pass: 1 sin/step, cvt two steps behind (pipelined)", w, out, sink);
    runpass<5>("pass: 1 value/step, sine = 7-VALU polynomial", w, out, sink);
    runpass<6>("pass: 2 values/2 steps, polynomial in v_pk_*_f32", w, out, sink);
    runpass<7>("pass: 2 values/2 steps, polynomial in v_pk_*_f16", w, out, sink);
    run<6>("builtin + ds_read_b128 per step", w, out, sink);
    return 0;
}
