// Host cost of enqueuing a frame's worth of kernels: seven plain launches against one hipGraphLaunch of the same seven
// (captured from the stream).  hipcc --offload-arch=gfx950 -O2 scripts/ubench/graph_launch.hip -o scripts/ubench/graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(float *p, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * s + 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float *p;
    const int n = 1 << 22;   // ~6 us of GPU work per kernel: the queue never runs dry, the host cost is what shows
    CK(hipMalloc(&p, n * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto frame = [&]() {
        for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, p, n, 1.0f + k);
    };
    for (int i = 0; i < 50; ++i) frame();
    CK(hipStreamSynchronize(st));
    const int reps = 2000;
    double t0 = now_us();
    for (int i = 0; i < reps; ++i) { frame(); CK(hipEventRecord(ev, st)); }
    double t1 = now_us();
    CK(hipStreamSynchronize(st));
    double t2 = now_us();
    printf("7 plain launches + event record: host %.2f us per frame (enqueue), %.2f us per frame incl. drain\n", (t1 - t0) / reps, (t2 - t0) / reps);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    frame();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = now_us();
    for (int i = 0; i < reps; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(ev, st)); }
    t1 = now_us();
    CK(hipStreamSynchronize(st));
    t2 = now_us();
    printf("hipGraphLaunch (7 kernel nodes) + event record: host %.2f us per frame (enqueue), %.2f us per frame incl. drain\n", (t1 - t0) / reps, (t2 - t0) / reps);
    // an event record INSIDE the capture, waited for by the host afterwards?
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, p, n, 1.0f + k);
    hipError_t er = hipEventRecord(ev, st);
    for (int k = 3; k < 7; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, p, n, 1.0f + k);
    hipError_t ec = hipStreamEndCapture(st, &g2);
    printf("event record inside a capture: record -> %s, end capture -> %s\n", hipGetErrorString(er), hipGetErrorString(ec));
    if (ec == hipSuccess && hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0) == hipSuccess) {
        hipError_t el = hipGraphLaunch(ge2, st);
        hipError_t es = hipEventSynchronize(ev);
        printf("   launch -> %s, host wait on that event -> %s\n", hipGetErrorString(el), hipGetErrorString(es));
    }
    (void)hipGetLastError();
    CK(hipStreamSynchronize(st));
    return 0;
}
