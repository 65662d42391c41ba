// heap_bench.hip -- cycles per heap_pop + heap_push on an LDS heap of k entries: sequential forms against ParHeap.
// build: hipcc --offload-arch=gfx950 -O3 -I gamma_amd/csrc tools/exp/heap_bench.hip -o tools/exp/heap_bench ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "heap_dev.h"
using namespace gh;

template <int MODE>
__global__ __launch_bounds__(64) void k_bench(int k, int n, const float* vals, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) uint2 h[1024 + 2];
    const int lane = threadIdx.x;
    heap_fill(h, k, lane, 64);
    __builtin_amdgcn_wave_barrier();
    float top = kHeapFltMax;
    int taken = 0;
    RegHeap<1> rh;
    rh.fill();
    const unsigned long long t0 = wall_clock64();
    for (int j0 = 0; j0 < n; j0 += 64) {
        const float dv = vals[j0 + lane];
        unsigned long long m = __ballot(top > dv);
        while (m) {
            const int l = (int)__ffsll((long long)m) - 1;
            const float val = hw_readlane_f(dv, l);
            if (MODE == 0) {
                heap_pop_seq(h, k);
                heap_push_seq(h, k, val, (unsigned)(j0 + l));
                top = hs_f(h[1].x);
            } else if (MODE == 1) {
                const float root = par_heap_pop(h, k);
                top = par_heap_push(h, k, val, (unsigned)(j0 + l)) ? val : root;
            } else if (MODE == 2) {   // the loop alone
                top = val + 1.0f;
            } else if (MODE == 3) {   // pop alone
                top = fminf(par_heap_pop(h, k), val + 1.0f);
            } else if (MODE == 4) {   // push alone
                top = par_heap_push(h, k, val, (unsigned)(j0 + l)) ? val + 1.0f : val + 1.0f;
            } else if (MODE == 5) {   // sequential pop alone
                heap_pop_seq(h, k);
                top = val + 1.0f;
            } else if (MODE == 6) {   // ParHeap<1> pop, no dispatch
                top = fminf(ParHeap<1>::pop(h, k), val + 1.0f);
            } else if (MODE == 7) {   // the heap in one register (k <= 63)
                rh.pop(k);
                rh.push(k, val, (unsigned)(j0 + l));
                top = rh.top();
            }
            taken++;
            const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
            m = __ballot(top > dv) & above;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) {
        out[0] = t1 - t0;
        out[1] = (unsigned long long)taken;
    }
    float acc = 0.f;
    for (int i = 1 + lane; i <= k; i += 64) acc += __uint_as_float(h[i].x) + (float)h[i].y;
    sink[lane] = acc;
}

int main(int argc, char** argv) {
    const int n = 1 << 13;   // descending values: every candidate is accepted (pure sift cost)
    float* hv = (float*)malloc(n * sizeof(float));
    srand(7);
    for (int i = 0; i < n; i++) hv[i] = (float)(2 * (n - i) + rand() % 2);
    float *dv, *sink;
    unsigned long long* out;
    hipMalloc((void**)&dv, n * sizeof(float));
    hipMalloc((void**)&sink, 64 * sizeof(float));
    hipMalloc((void**)&out, 16);
    hipMemcpy(dv, hv, n * sizeof(float), hipMemcpyHostToDevice);
    const char* names[] = {"seq pop+push", "par pop+push", "loop alone", "par pop", "par push", "seq pop", "ParHeap<1> pop", "RegHeap<1> pop+push"};
    for (int k : {10, 32, 63, 100, 200, 256}) {
        for (int mode = 0; mode < 8; mode++) {
            if (mode == 6 && k > 128) continue;
            if (mode == 7 && k > 63) continue;
            if (mode >= 2 && mode <= 6 && k != 100) continue;
            unsigned long long ho[2];
            for (int rep = 0; rep < 2; rep++) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 1: hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 2: hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 3: hipLaunchKernelGGL(k_bench<3>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 4: hipLaunchKernelGGL(k_bench<4>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 5: hipLaunchKernelGGL(k_bench<5>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    case 6: hipLaunchKernelGGL(k_bench<6>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                    default: hipLaunchKernelGGL(k_bench<7>, dim3(1), dim3(64), 0, 0, k, n, dv, out, sink); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
            printf("k %4d %-20s: %llu accepted, %.1f ns each (100 MHz clock)\n", k, names[mode], ho[1], 10.0 * (double)ho[0] / (double)ho[1]);
        }
    }
    return 0;
}
