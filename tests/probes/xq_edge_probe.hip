// Hardware probe (round 4): what does a cross-queue dependency cost on this stack, by mechanism -- and which mechanisms
// survive HIP-graph capture?  (fork the second queue once per decode STEP and carry the per-layer
// dependencies through memory flags.)
// Per "layer": stream A runs pre (5 us) -> [A: suffix (20 us) || B: prefix (15 us)] -> A: merge (3 us); 32 layers per step.
//   serial   one stream: pre, prefix, suffix, merge                                 (43 us of kernels per layer)
//   events   fork / join with hipEventRecord + hipStreamWaitEvent per layer          (28 us on the critical path)
//   memops   fork / join with hipStreamWriteValue32 + hipStreamWaitValue32 on one flag word per direction
//   graph    the same calls captured into one hipGraph (B joins the capture once per step) and replayed
// The kernels are one-wave spinners on the 100 MHz wall clock and stamp their start / end: nothing competes for CUs, the
// difference between a row and 28 us x 32 is synchronisation, and the stamps say whether the edges held.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

static const int L = 32;
__device__ unsigned long long stamps[2 * 4 * L];
__global__ void spin(int us, int slot) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = wall_clock64(); }
}

static hipStream_t A, B;
static hipEvent_t ef[L + 1], ej[L + 1];
static unsigned *flagF, *flagJ;  // signal memory
static unsigned epoch = 0;
static bool ring_epoch = false;  // graph: the values are baked into the nodes, the flags are reset at the head of a replay

static void layer_serial(int l) { spin<<<1, 64, 0, A>>>(5, 4 * l); spin<<<1, 64, 0, A>>>(15, 4 * l + 1); spin<<<1, 64, 0, A>>>(20, 4 * l + 2); spin<<<1, 64, 0, A>>>(3, 4 * l + 3); }
static void layer_events(int l) {
    spin<<<1, 64, 0, A>>>(5, 4 * l);
    CK(hipEventRecord(ef[l], A)); CK(hipStreamWaitEvent(B, ef[l], 0));
    spin<<<1, 64, 0, B>>>(15, 4 * l + 1);
    spin<<<1, 64, 0, A>>>(20, 4 * l + 2);
    CK(hipEventRecord(ej[l], B)); CK(hipStreamWaitEvent(A, ej[l], 0));
    spin<<<1, 64, 0, A>>>(3, 4 * l + 3);
}
static void layer_memops(int l) {
    epoch = ring_epoch ? (unsigned)l + 1u : epoch + 1u;
    spin<<<1, 64, 0, A>>>(5, 4 * l);
    CK(hipStreamWriteValue32(A, flagF, epoch, 0)); CK(hipStreamWaitValue32(B, flagF, epoch, hipStreamWaitValueGte, 0xffffffffu));
    spin<<<1, 64, 0, B>>>(15, 4 * l + 1);
    spin<<<1, 64, 0, A>>>(20, 4 * l + 2);
    CK(hipStreamWriteValue32(B, flagJ, epoch, 0)); CK(hipStreamWaitValue32(A, flagJ, epoch, hipStreamWaitValueGte, 0xffffffffu));
    spin<<<1, 64, 0, A>>>(3, 4 * l + 3);
}

// did the edges hold in the LAST pass?  prefix(l) starts after pre(l) ends, merge(l) after prefix(l) and suffix(l) end
static void verify(const char* what, double us) {
    unsigned long long st[2 * 4 * L]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(stamps), sizeof(st)));
    int bad = 0; double ov = 0;
    for (int l = 0; l < L; ++l) {
        const unsigned long long *pre = st + 8 * l, *pf = pre + 2, *sf = pre + 4, *mg = pre + 6;
        bad += pf[0] < pre[1]; bad += mg[0] < pf[1]; bad += mg[0] < sf[1];
        const unsigned long long lo = pf[0] > sf[0] ? pf[0] : sf[0], hi = pf[1] < sf[1] ? pf[1] : sf[1];
        ov += hi > lo ? (double)(hi - lo) / 100.0 : 0.0;
    }
    printf("%-14s %7.2f us / layer | broken edges %2d of %d | prefix || suffix overlap %4.1f us / layer (15 possible)\n", what, us, bad, 3 * L, ov / L);
}

template <class F>
static void eager(F f, int steps, const char* what) {
    for (int l = 0; l < L; ++l) f(l);
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) for (int l = 0; l < L; ++l) f(l);
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    verify(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps / L);
}
template <class F>
static void graphed(F f, int steps, const char* what) {
    hipGraph_t g; hipGraphExec_t ge;
    hipError_t e = hipStreamBeginCapture(A, hipStreamCaptureModeGlobal);
    if (e != hipSuccess) { printf("%s: begin capture: %s\n", what, hipGetErrorString(e)); return; }
    if (ring_epoch) { CK(hipStreamWriteValue32(A, flagF, 0, 0)); CK(hipStreamWriteValue32(A, flagJ, 0, 0)); }
    CK(hipEventRecord(ef[L], A)); CK(hipStreamWaitEvent(B, ef[L], 0));  // B joins the capture ONCE per step
    for (int l = 0; l < L; ++l) f(l);
    CK(hipEventRecord(ej[L], B)); CK(hipStreamWaitEvent(A, ej[L], 0));
    e = hipStreamEndCapture(A, &g);
    if (e != hipSuccess) { printf("%s: end capture: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { printf("%s: instantiate: %s\n", what, hipGetErrorString(e)); return; }
    for (int s = 0; s < 3; ++s) CK(hipGraphLaunch(ge, A));
    CK(hipStreamSynchronize(A));
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) CK(hipGraphLaunch(ge, A));
    CK(hipStreamSynchronize(A));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps / L;
    printf("(%s: %zu graph nodes for %d kernels) ", what, nn, 4 * L);
    verify(what, us);
}

int main() {
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    for (int l = 0; l <= L; ++l) { CK(hipEventCreateWithFlags(&ef[l], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[l], hipEventDisableTiming)); }
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d; kernels per layer: 43 us serial, 28 us on the critical path when prefix || suffix\n", can);
    CK(hipExtMallocWithFlags((void**)&flagF, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&flagJ, 8, hipMallocSignalMemory));
    CK(hipMemset(flagF, 0, 8)); CK(hipMemset(flagJ, 0, 8)); CK(hipDeviceSynchronize());
    eager(layer_serial, 5, "eager serial");
    eager(layer_events, 5, "eager events");
    if (can) eager(layer_memops, 5, "eager memops");
    CK(hipDeviceSynchronize());
    graphed(layer_serial, 10, "graph serial");
    graphed(layer_events, 10, "graph events");
    if (can) { ring_epoch = true; CK(hipMemset(flagF, 0, 8)); CK(hipMemset(flagJ, 0, 8)); CK(hipDeviceSynchronize()); graphed(layer_memops, 10, "graph memops"); }
    // the graph API's own node for this (hipGraphAddBatchMemOpNode) needs hipStreamBatchMemOp semantics; the stream form:
    hipStreamBatchMemOpParams q; memset(&q, 0, sizeof(q));
    q.writeValue.operation = hipStreamMemOpWriteValue32; q.writeValue.address = (hipDeviceptr_t)flagF; q.writeValue.value = 7;
    printf("hipStreamBatchMemOp(write value): %s\n", hipGetErrorString(hipStreamBatchMemOp(A, 1, &q, 0)));
    return 0;
}
