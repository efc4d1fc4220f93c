// Micro-benchmark: what does the vector L1 (TCP) charge for a wave-level global_load_dwordx4 whose data is already in L1?
// The C3 march kernel is bound by this unit (0.94 TCP_TOTAL_CACHE_ACCESSES per clock per CU); this
// establishes the ceiling that number is measured against (VERDICT r2 #6a) and how the charge depends on the number
// of 128-B lines the four lanes of a quad touch.
//
// Every wave re-reads its own small window (<= 8 KiB; 16 waves per CU -> the working set stays far inside L1 / L2), eight
// independent loads per trip so that latency is hidden, results folded into a checksum.  Patterns (per wave-level load):
//   same     all 64 lanes the same 16 B                            1 line
//   linear   lane l reads 16 B at 16 l            (1 KiB contiguous: 8 lines; a quad = 64 B inside one line)
//   quad2    a quad's lanes split over 2 lines    (lanes 2k, 2k+1 share a line)
//   quad4    every lane its own line              (64 lines)
//   dword    lane l reads 4 B at 4 l              (256 B contiguous: 2 lines) — the dword gather of round 1's kernels
//   q2slots / q4slots / q4slot8 / q1even / q1mix: a quad over 2 / 4 lines or one line, with the four lanes on DIFFERENT 16-B slots
//            of their lines — separates "how many lines" from "which 16-B bank" (quad2 / quad4 put every lane on slot 0 / 1)
//   march    the C3 kernel's own shape: 2x2-pixel quads 0.65 voxels apart in a 4x2x1 brick grid (computed addresses)
// Output: ns per wave-level load per CU and, with the shader clock, loads per clock per CU.  Run it under
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -- /tmp/tcp_tag_rate
// for the look-ups per load (tools/micro/run_tcp_tag_rate.sh does both and writes profiles/r03_tcp_tag_rate.txt).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/tcp_tag_rate.hip -o /tmp/tcp_tag_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void tag_kernel(const char* __restrict__ buf, float* __restrict__ sink, int trips,
                                                  unsigned long long* __restrict__ clocks) {
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const char* base = buf + (size_t)(wave & 1023u) * 8192;          // this wave's 8-KiB window
    unsigned off;
    if (MODE == 0) off = 0;
    else if (MODE == 1) off = lane * 16;
    else if (MODE == 2) off = (lane >> 1) * 128 + (lane & 1) * 16;
    else if (MODE == 3) off = lane * 128;
    else if (MODE == 4) off = lane * 4;
    else if (MODE == 6) off = (lane >> 1) * 128 + (lane & 3) * 16;                    // quad over 2 lines, four different 16-B slots
    else if (MODE == 7) off = lane * 128 + (lane & 3) * 16;                           // quad over 4 lines, four different slots
    else if (MODE == 8) off = lane * 128 + (lane & 7) * 16;                           // every lane its own line, 8 slots in turn
    else if (MODE == 9) off = (lane >> 2) * 128 + (lane & 3) * 32;                    // one line per quad, slots 0 2 4 6
    else if (MODE == 10) off = (lane >> 2) * 128 + (lane & 3) * 64 % 128 + ((lane & 3) >> 1) * 16;   // one line per quad, slots 0 4 1 5
    else {
        // 8x8 pixel packet in Morton order, pixels 0.65 voxels apart, sampled in a grid of 4 x 2 x 1 bricks of 16-B voxels
        const unsigned px = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), py = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
        const unsigned x = (unsigned)(0.37f + 0.65f * px), y = (unsigned)(0.81f + 0.65f * py);       // voxel coordinates (0..5)
        off = ((y >> 1) * 2 + (x >> 2)) * 128 + ((y & 1) * 4 + (x & 3)) * 16;
    }
    f4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned o = MODE == 4 ? off + k * 256 : (MODE == 3 || MODE >= 7 && MODE <= 8) ? off : (MODE == 2 || MODE == 6) ? ((off + k * 4096) & 8191u) : off + k * 1024;
            if (MODE == 4) { v[k] = (f4){ *reinterpret_cast<const float*>(base + (o & 8191u)), 0.0f, 0.0f, 0.0f }; }
            else v[k] = *reinterpret_cast<const f4*>(base + (o & 8191u));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
        asm volatile("" : "+v"(off));                                 // the address is re-read every trip: no hoisting of the loads
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
    if (lane == 0 && wave == 0) clocks[MODE] = t1 - t0;
}

int main(int argc, char** argv) {
    const int trips = argc > 1 ? atoi(argv[1]) : 4096;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 4;      // 16 waves per CU
    char* buf; float* sink; unsigned long long* clocks;
    hipMalloc(&buf, 1024 * 8192 + 4096);
    hipMemset(buf, 0, 1024 * 8192 + 4096);
    hipMalloc(&sink, 64);
    hipMalloc(&clocks, 128);
    hipMemset(clocks, 0, 128);
    const char* names[11] = { "same", "linear", "quad2", "quad4", "dword", "march", "q2slots", "q4slots", "q4slot8", "q1even", "q1mix" };
    void (*kern[11])(const char*, float*, int, unsigned long long*) = { tag_kernel<0>, tag_kernel<1>, tag_kernel<2>, tag_kernel<3>, tag_kernel<4>, tag_kernel<5>,
                                                                       tag_kernel<6>, tag_kernel<7>, tag_kernel<8>, tag_kernel<9>, tag_kernel<10> };
    printf("# %s, %d CUs, clockRate %d kHz; %d blocks x 256 threads (16 waves per CU), %d trips x 8 loads per wave\n",
           prop.name, cus, prop.clockRate, blocks, trips);
    printf("# pattern   ms      ns per wave-level load per CU   clk per load per CU (at clockRate)   s_memtime ticks of wave 0\n");
    for (int m = 0; m < 11; ++m) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(256), 0, 0, buf, sink, 64, clocks);      // warm the caches
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(256), 0, 0, buf, sink, trips, clocks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long ticks[16];
        hipMemcpy(ticks, clocks, 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const double loadsPerCu = 16.0 * trips * 8.0;
        const double ns = ms * 1e6 / loadsPerCu;
        printf("%-8s %8.3f   %10.3f                      %8.2f                             %llu\n", names[m], ms, ns,
               ns * prop.clockRate * 1e-6, ticks[m]);
    }
    return 0;
}
