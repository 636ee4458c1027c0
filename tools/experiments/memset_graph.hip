// Reproducer for the round-1 finding "a captured hipMemsetAsync of a multiple of 16 bytes came back with garbage from
// the second replay on" (ROCm 7.2, gfx950).  Stream-captures {memset(buf, 0, n); probe kernel: out[rep] |= buf != 0,
// then dirty buf} and replays the graph several times; prints, per size, in which replays the probe saw non-zero bytes.
// Build: hipcc --offload-arch=gfx950 -O2 tools/experiments/memset_graph.hip -o tools/experiments/build/memset_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void probe(unsigned char *buf, int n, int *seen, const int *rep)
{
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        bad |= buf[i] != 0;
    if (bad)
        atomicOr(&seen[*rep], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        buf[i] = 0xA5; // dirty it again: the next replay's memset must clear it
}
__global__ void bump(int *rep) { *rep += 1; }

int main()
{
    const int sizes[] = {4, 12, 16, 32, 48, 64, 256, 1024, 4096};
    int fails = 0;
    for (int offset : {0, 256}) // a buffer of its own, and a sub-allocation at an offset inside a bigger block
    for (int n : sizes) {
        unsigned char *base;
        int *seen, *rep;
        CK(hipMalloc(&base, 1 << 20));
        unsigned char *buf = base + offset;
        CK(hipMalloc(&seen, 64 * sizeof(int)));
        CK(hipMalloc(&rep, sizeof(int)));
        CK(hipMemset(base, 0xA5, 1 << 20));
        CK(hipMemset(seen, 0, 64 * sizeof(int)));
        CK(hipMemset(rep, 0, sizeof(int)));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(buf, 0, n, st));
        probe<<<1, 256, 0, st>>>(buf, n, seen, rep);
        bump<<<1, 1, 0, st>>>(rep);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 8; ++r)
            CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        int h[64];
        CK(hipMemcpy(h, seen, sizeof(h), hipMemcpyDeviceToHost));
        printf("offset %3d size %5d: dirty seen in replays:", offset, n);
        int any = 0;
        for (int r = 0; r < 8; ++r)
            if (h[r]) { printf(" %d", r); any = 1; }
        printf(any ? "  <-- memset node did not clear\n" : " none\n");
        fails += any;
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipStreamDestroy(st));
        CK(hipFree(base)); CK(hipFree(seen)); CK(hipFree(rep));
    }
    printf("RESULT: %d of 18 cases failed\n", fails);
    return 0;
}
