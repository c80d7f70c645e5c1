// clock_probe.hip -- how fast does the shader clock run during SHORT kernels on an otherwise idle device?
// One wave executes a chain of dependent v_fma_f32 (4 issue cycles each on wave64 CDNA) and samples the 100 MHz wall clock every 4096 links:
// ns per link * (1 / 4 cycles) = effective shader clock.  Run as: hipcc --offload-arch=gfx950 -O3 scripts/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>

__global__ void k_chain(float *out, long long *stamps, int n_slices) {
    float x = out[threadIdx.x];
    for (int s = 0; s < n_slices; ++s) {
        if (threadIdx.x == 0) stamps[s] = wall_clock64();
#pragma unroll 64
        for (int i = 0; i < 4096; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    }
    if (threadIdx.x == 0) stamps[n_slices] = wall_clock64();
    out[threadIdx.x] = x;
}

int main() {
    float *d; long long *st;
    const int n_slices = 256;
    hipMalloc(&d, 256); hipMemset(d, 0, 256);
    hipHostMalloc(&st, sizeof(long long) * (n_slices + 1));
    auto run = [&](const char *what, int slices) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, st, slices);
        hipDeviceSynchronize();
        double first = (st[1] - st[0]) * 10.0 / 4096, last = (st[slices] - st[slices - 1]) * 10.0 / 4096, all = (st[slices] - st[0]) * 10.0 / (4096.0 * slices);
        printf("%-46s %4d slices: ns per dependent fma  first %.3f  last %.3f  mean %.3f  => clock (4 cycles per fma) first %.0f MHz, last %.0f MHz\n", what, slices, first, last, all,
               4000.0 / first, 4000.0 / last);
    };
    run("cold (first launch)", 256);
    run("right after", 256);
    usleep(200000);
    run("after 200 ms idle", 4);
    run("right after (short)", 4);
    for (int i = 0; i < 5; ++i) { usleep(100); run("short kernel, 100 us pause before", 8); }
    for (int i = 0; i < 3; ++i) { usleep(2000); run("short kernel, 2 ms pause before", 8); }
    // a burst of short kernels back to back (what an RL loop looks like): 2000 launches of ~25 us
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, st, 2);
        hipDeviceSynchronize();
        run("after a burst of 2000 short kernels", 4);
    }
    return 0;
}
