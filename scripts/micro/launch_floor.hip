// Micro-benchmark: what a dependent launch costs on one stream by grid shape (empty kernels that read one argument),
// the way the level-by-level search chains its launches.  Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
struct Big { int v[64]; };   // a kernel argument block of 256 bytes, like the search kernels'
__global__ void k_empty(Big b, int *out, int stop) { if (stop == 12345) out[0] = b.v[3]; }
__global__ void k_touch(Big b, float *buf, int n_per_block)   // every block writes n_per_block floats (dirty lines at the launch's end)
{
    float *p = buf + (size_t)blockIdx.x * n_per_block;
    for (int i = threadIdx.x; i < n_per_block; i += blockDim.x) p[i] = (float)b.v[1];
}
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    int *out; float *buf;
    CHK(hipMalloc(&out, 4096));
    CHK(hipMalloc(&buf, 64u << 20));
    CHK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t st; CHK(hipStreamCreate(&st));
    Big b{};
    struct Shape { int blocks, threads, lds; };
    std::vector<Shape> shapes = {{256, 512, 0}, {256, 512, 33 * 1024}, {416, 512, 33 * 1024}, {256, 256, 0}, {1024, 64, 0}, {256, 64, 0},
                                 {64, 64, 0}, {1, 64, 0}, {256, 1024, 0}, {512, 512, 33 * 1024}, {832, 256, 33 * 1024}};
    const int N = 3000;
    for (const Shape &s : shapes) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(s.blocks), dim3(s.threads), s.lds, st, b, out, i);
            CHK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("empty  blocks %4d threads %4d lds %6d : %.2f us per launch\n", s.blocks, s.threads, s.lds, us);
        }
    }
    // dirty data left by the launch before: 256 blocks x 512 threads writing 0 / 4 / 16 / 64 KB each, an empty launch between
    for (int kb : {0, 4, 16, 64}) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, st, b, buf, kb * 256);
                hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, b, out, i);
            }
            CHK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("pair   writer of %2d KB per block + empty : %.2f us per pair\n", kb, us);
        }
    }
    // the same chain without the host: N kernel nodes captured into one graph, replayed (device-side cost per dependent launch)
    for (const Shape &s : {Shape{256, 512, 33 * 1024}, Shape{416, 512, 33 * 1024}, Shape{256, 256, 0}, Shape{1, 64, 0}}) {
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 600; i++) hipLaunchKernelGGL(k_empty, dim3(s.blocks), dim3(s.threads), s.lds, st, b, out, i);
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 5; k++) CHK(hipGraphLaunch(ge, st));
            CHK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 3000;
            if (rep == 2) printf("graph  blocks %4d threads %4d lds %6d : %.2f us per kernel node\n", s.blocks, s.threads, s.lds, us);
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
