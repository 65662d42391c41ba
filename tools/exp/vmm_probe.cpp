// probe of the HIP virtual-memory-management calls on this runtime: which interleavings of hipMemMap / hipMemSetAccess with
// ordinary allocations, kernels and threads work.  build: hipcc --offload-arch=gfx950 -O1 -o vmm_probe vmm_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <thread>
#include <atomic>
#include <vector>
__global__ void k_touch(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
static char* va;
static size_t gran, mapped;
static int map_more(size_t add, const char* tag) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, add, &prop, 0);
    if (e != hipSuccess) { printf("%s: create %s\n", tag, hipGetErrorString(e)); return 1; }
    e = hipMemMap(va + mapped, add, 0, h, 0);
    if (e != hipSuccess) { printf("%s: map %s\n", tag, hipGetErrorString(e)); return 1; }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va + mapped, add, &acc, 1);
    if (e != hipSuccess) {
        printf("%s: setaccess(chunk) %s", tag, hipGetErrorString(e));
        (void)hipGetLastError();
        e = hipMemSetAccess(va, mapped + add, &acc, 1);
        printf(" | setaccess(whole range from base) %s\n", hipGetErrorString(e));
        if (e != hipSuccess) return 1;
    }
    mapped += add;
    printf("%s: ok, mapped %zu MB\n", tag, mapped >> 20);
    return 0;
}
int main() {
    hipSetDevice(0);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    size_t gmin = 0;
    hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    size_t fr, tot;
    hipMemGetInfo(&fr, &tot);
    size_t want = (tot + gran - 1) / gran * gran;
    hipError_t e = hipMemAddressReserve((void**)&va, want, 0, nullptr, 0);
    printf("gran %zu min %zu reserve %zu GB: %s va=%p\n", gran, gmin, want >> 30, hipGetErrorString(e), va);
    const size_t C = 64 << 20;
    map_more(C, "A1 plain");
    map_more(C, "A2 plain");
    void* other = nullptr;
    hipMalloc(&other, 256 << 20);
    map_more(C, "B after hipMalloc");
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, (float*)va, (size_t)1 << 18);
    printf("kernel on the mapped range: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    map_more(C, "C after a kernel on the range (device idle)");
    hipStream_t s;
    hipStreamCreate(&s);
    hipLaunchKernelGGL(k_touch, dim3(1 << 20), dim3(256), 0, s, (float*)other, (size_t)64 << 20);
    map_more(C, "D beside a kernel on another allocation");
    hipStreamSynchronize(s);
    hipLaunchKernelGGL(k_touch, dim3(1 << 18), dim3(256), 0, s, (float*)va, (size_t)64 << 20);
    map_more(C, "E beside a kernel on the range");
    hipStreamSynchronize(s);
    std::atomic<bool> stop{false};
    std::thread t([&] {
        hipSetDevice(0);
        hipStream_t s2;
        hipStreamCreate(&s2);
        while (!stop) {
            void* p = nullptr;
            hipMalloc(&p, 8 << 20);
            hipLaunchKernelGGL(k_touch, dim3(1 << 12), dim3(256), 0, s2, (float*)va, (size_t)1 << 20);
            hipStreamSynchronize(s2);
            hipFree(p);
        }
    });
    for (int i = 0; i < 6; i++) map_more(C, "F beside a thread allocating and launching");
    stop = true;
    t.join();
    map_more(3 * C, "G a larger chunk");
    map_more(C + gran, "H an odd chunk");
    printf("final: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
