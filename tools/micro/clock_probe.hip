// Calibration probe: core clock vs the 100 MHz wall clock with ONE busy workgroup; LDS / L2 / dependent-FMA latency.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* g, int* chase, long long* out, int n) {
    __shared__ float lds[1024];
    const int t = threadIdx.x;
    lds[t] = (float) t;
    __syncthreads();
    float x = 1.0f + t * 1e-9f;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) x = x * 1.0000001f + 1e-9f;          // dependent mul+add (no contraction)
    long long w1 = wall_clock64(), c1 = clock64();
    int idx = t & 1023;
    for (int i = 0; i < n; ++i) idx = (int) lds[idx] & 1023;          // dependent LDS reads
    long long w2 = wall_clock64();
    int p = t;
    for (int i = 0; i < 4096; ++i) p = chase[p];                      // dependent global loads (L2 resident ring)
    long long w3 = wall_clock64();
    if (t == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = w2 - w1; out[3] = w3 - w2; }
    g[t] = x + idx + p;
}
int main() {
    const int n = 100000;
    float* g; int* chase; long long* out;
    hipMalloc(&g, 4096); hipMalloc(&chase, 65536 * 4); hipMalloc(&out, 64);
    int* h = new int[65536];
    for (int i = 0; i < 65536; ++i) h[i] = (i + 64 * 17) & 65535;      // stride of 17 cache lines
    hipMemcpy(chase, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, chase, out, n);
        hipDeviceSynchronize();
        long long o[4]; hipMemcpy(o, out, 32, hipMemcpyDeviceToHost);
        printf("fma loop: %lld wall ticks (100MHz) %lld core clocks -> core %.0f MHz, %.2f core clk per dependent mul+add pair | "
               "LDS dependent read %.1f ns | global dependent load %.1f ns\n", o[0], o[1], 100.0 * o[1] / o[0], (double) o[1] / n,
               o[2] * 10.0 / n, o[3] * 10.0 / 4096);
    }
    return 0;
}
