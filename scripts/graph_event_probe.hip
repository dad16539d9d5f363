// probe: are hipEventRecord calls captured into a hipGraph, and does hipEventElapsedTime work on them after a replay?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
int main() {
    float* d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreate(&e2);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipError_t r = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    printf("begin capture: %s\n", hipGetErrorString(r));
    r = hipEventRecord(e0, s);
    printf("record e0 in capture: %s\n", hipGetErrorString(r));
    spin<<<1, 64, 0, s>>>(d, 200000);
    r = hipEventRecord(e1, s);
    printf("record e1 in capture: %s\n", hipGetErrorString(r));
    spin<<<1, 64, 0, s>>>(d, 400000);
    r = hipEventRecord(e2, s);
    r = hipStreamEndCapture(s, &g);
    printf("end capture: %s\n", hipGetErrorString(r));
    r = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("instantiate: %s\n", hipGetErrorString(r));
    for (int i = 0; i < 3; ++i) {
        r = hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        float a = -1, b = -1;
        hipError_t ra = hipEventElapsedTime(&a, e0, e1), rb = hipEventElapsedTime(&b, e1, e2);
        printf("replay %d: launch %s; e0->e1 %.3f ms (%s), e1->e2 %.3f ms (%s)\n", i, hipGetErrorString(r), a, hipGetErrorString(ra), b, hipGetErrorString(rb));
    }
    return 0;
}
