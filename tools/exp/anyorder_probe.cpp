// hipcc --offload-arch=gfx950 -O2 tools/exp/anyorder_probe.cpp -o /tmp/anyorder_probe && /tmp/anyorder_probe
// Does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) let a kernel start beside the previous one of the SAME stream on
// this part?  (hip_ext.h says the flag is not supported on GFX9xx.)  A = 4 workgroups spinning ~200 us, B = a short kernel;
// in order: t(A + B) ~ t(A) + t(B); any order honoured: B's start (its own event pair) falls inside A.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
__global__ void fill(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 1.f;
}
int main() {
    int* d_o; float* d_p; const int n = 64 << 20;
    hipMalloc(&d_o, 1024); hipMalloc(&d_p, (size_t)n * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1, b0, b1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&b0); hipEventCreate(&b1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s, 20000LL, d_o);   // wall_clock64: 100 MHz -> 200 us
            hipExtLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, s, b0, b1, mode ? hipExtAnyOrderLaunch : 0, d_p, n);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float total = 0, fb = 0, gap = 0;
            hipEventElapsedTime(&total, e0, e1); hipEventElapsedTime(&fb, b0, b1); hipEventElapsedTime(&gap, e0, b0);
            printf("mode %s: total %.1f us, fill %.1f us, fill starts %.1f us after the stream's start\n",
                   mode ? "any-order" : "in-order ", total * 1e3, fb * 1e3, gap * 1e3);
        }
    }
    return 0;
}
