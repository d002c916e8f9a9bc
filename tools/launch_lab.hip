// Host-side cost of the HIP calls the training step makes (development tool): kernel launch with small / large by-value
// arguments, event record, cross-stream wait.  hipcc --offload-arch=gfx950 -O3 tools/launch_lab.hip -o launch_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[400]; };
struct Huge { char b[3300]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_big(Big a, int* p) { if (p && threadIdx.x == 999) *p = a.b[0]; }
__global__ void k_huge(Huge a, int* p) { if (p && threadIdx.x == 999) *p = a.b[0]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev[64]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    Big big = {}; Huge huge = {};
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, (int*)nullptr);
        double t1 = now(); hipStreamSynchronize(s);
        double t2 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(256), dim3(768), 0, s, big, (int*)nullptr);
        double t3 = now(); hipStreamSynchronize(s);
        double t4 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_huge, dim3(256), dim3(768), 0, s, huge, (int*)nullptr);
        double t5 = now(); hipStreamSynchronize(s);
        double t6 = now();
        for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], s); hipStreamWaitEvent(s2, ev[i & 63], 0); }
        double t7 = now(); hipDeviceSynchronize();
        if (rep) printf("host us per call: launch(8 B args) %.2f | launch(400 B) %.2f | launch(3.3 KB) %.2f | record+wait %.2f   (GPU drain: %.0f %.0f us)\n",
                        (t1 - t0) / N, (t3 - t2) / N, (t5 - t4) / N, (t7 - t6) / N, t2 - t1, t4 - t3);
    }
    return 0;
}
