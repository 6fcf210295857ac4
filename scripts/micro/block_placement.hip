// Micro-benchmark: where the workgroups of a (tiles x blocks) grid of 512-thread workgroups with 33 KB of LDS land -
// XCD, shader engine, compute unit of every workgroup, all resident at once (each spins 20 us) - to see which blocks
// of the level kernel share a compute unit.  Build: hipcc --offload-arch=gfx950 -O3 block_placement.hip -o block_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k_where(unsigned *out)
{
    extern __shared__ float pad[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);   // HW_REG_XCC_ID
        out[blockIdx.y * gridDim.x + blockIdx.x] = (xcc & 15) << 16 | (hw & 0xffff);
        pad[0] = 1.0f;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}   // 100 MHz ticks: 20 us
}
int main(int argc, char **argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 32, blocks = argc > 2 ? atoi(argv[2]) : 13;
    unsigned *d; std::vector<unsigned> h(tiles * blocks);
    hipMalloc(&d, h.size() * 4);
    hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_where, dim3(tiles, blocks), dim3(512), 33 * 1024, 0, d);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // (xcc, se, sh, cu) -> linear block indices
    for (int b = 0; b < (int)h.size(); b++) {
        const unsigned v = h[b], key = (v >> 16) << 16 | (v & 0xff00);   // drop wave, simd, pipe
        cu[key].push_back(b);
    }
    printf("%d x %d workgroups on %zu compute units\n", tiles, blocks, cu.size());
    std::map<size_t, int> hist;
    for (auto &kv : cu) hist[kv.second.size()]++;
    for (auto &kv : hist) printf("  %d compute units hold %zu workgroups\n", kv.second, kv.first);
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 24) break;
        printf("  xcc %u hw %04x :", kv.first >> 16, kv.first & 0xffff);
        for (int b : kv.second) printf(" (tile %d, block %d)", b % tiles, b / tiles);
        printf("\n");
    }
    // does linear index mod 8 give the XCD?
    int same = 0;
    for (int b = 0; b < (int)h.size(); b++) same += ((h[b] >> 16) == (h[0] >> 16)) == (b % 8 == 0);
    printf("blocks whose XCD follows (linear index mod 8): %d of %zu\n", same, h.size());
    return 0;
}
