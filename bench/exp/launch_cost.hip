// Experiment: what does a kernel of a given geometry cost when it does (almost) nothing?
// 4 dependent kernels per "update", 100 updates captured in one graph, timed with events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int touch_lds) {
    extern __shared__ float sm[];
    if (touch_lds) sm[threadIdx.x] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = touch_lds ? sm[0] : 1.f;
}
static float run(int grid, int block, int lds, float* out, hipStream_t s) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, out, 1);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipEventRecord(e0, s);
    for (int w = 0; w < 10; ++w) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / 4000.f;
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    struct { int grid, block, lds; } cfg[] = {
        {201, 1024, 66 * 1024}, {201, 1024, 1024}, {201, 512, 33 * 1024}, {201, 256, 17 * 1024}, {201, 256, 1024},
        {804, 256, 17 * 1024}, {65, 768, 70 * 1024}, {65, 768, 1024}, {16, 768, 70 * 1024}, {65, 512, 70 * 1024},
        {256, 1024, 66 * 1024}, {1, 64, 1024}};
    for (auto c : cfg) printf("grid %4d block %4d lds %6d : %.2f us per kernel (back to back in a graph)\n", c.grid, c.block, c.lds, run(c.grid, c.block, c.lds, out, s));
    return 0;
}
