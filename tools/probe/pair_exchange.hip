// What would it cost to run ONE board's tower on TWO CUs?  (a one-board launch of a 128-channel net is
// matrix-pipe time on one CU -- 23.5 us per layer, 476 us per 9 x 128 launch -- while 156 of 256 CUs idle at 100 games.)
// Splitting the output channels over two workgroups halves the pipe time per layer but every layer then ends with an exchange:
// each workgroup publishes its half of the layer's output (48 rows x 64 channels fp32 = 12 KB), waits for the partner's flag and
// reads the partner's half.  This probe measures that exchange alone: P pairs, 19 "layers", no arithmetic.
//   partner = wg ^ 8: same XCD if workgroups are dealt round-robin over the 8 XCDs; wg ^ 1: neighbouring XCDs.
// Every wait is bounded (a pair that never meets sets *fail and leaves).
//   hipcc --offload-arch=gfx950 -O3 -o pair_exchange pair_exchange.hip && ./pair_exchange
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int LAYERS = 19, HALF = 48 * 64, NT = 512;      // floats per half

template <int MODE>      // 0: device-scope fences (what the memory model asks for); 1: the same without the data fence (lower bound, same-XCD only)
__global__ __launch_bounds__(NT) void k_pair(float *xbuf, unsigned *flags, unsigned long long *out, int pxor, unsigned *fail, float *sink) {
    __shared__ __attribute__((aligned(16))) float act[2 * HALF];
    const int wg = blockIdx.x, partner = wg ^ pxor, tid = threadIdx.x;
    for (int i = tid; i < 2 * HALF; i += NT) act[i] = (float)(wg + i);
    __syncthreads();
    unsigned long long t_ex = 0;
    const unsigned long long t0 = wall_clock64();
    bool dead = false;
    for (int layer = 0; layer < LAYERS && !dead; ++layer) {
        const unsigned long long ta = wall_clock64();
        float4 *dst = reinterpret_cast<float4 *>(xbuf + ((size_t)wg * 2 + (layer & 1)) * HALF);
        for (int i = tid; i < HALF / 4; i += NT) dst[i] = reinterpret_cast<const float4 *>(act)[i];
        if (MODE == 0) __threadfence();
        __syncthreads();
        __shared__ int s_dead;
        if (tid == 0) {
            s_dead = 0;
            __hip_atomic_store(&flags[wg], (unsigned)(layer + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&flags[partner], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(layer + 1)) {
                if (++spins > 400000) { *fail = 1; s_dead = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        dead = s_dead != 0;
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float4 *src = reinterpret_cast<const float4 *>(xbuf + ((size_t)partner * 2 + (layer & 1)) * HALF);
        for (int i = tid; i < HALF / 4; i += NT) reinterpret_cast<f4v *>(act + HALF)[i] = MODE == 0 ? reinterpret_cast<const f4v *>(src)[i] : __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src) + i);
        __syncthreads();
        t_ex += wall_clock64() - ta;
        // fold the partner's half into ours so that the next layer's data depends on it
        for (int i = tid; i < HALF; i += NT) act[i] = act[i] * 0.5f + act[HALF + i] * 0.25f;
        __syncthreads();
    }
    if (tid == 0) { out[2 * wg] = t_ex; out[2 * wg + 1] = wall_clock64() - t0; }
    if (tid == 0) sink[wg] = act[7];
}

template <int MODE>
static void run(const char *what, int pairs, int pxor) {
    const int wgs = 2 * pairs;
    float *xbuf, *sink; unsigned *flags, *fail; unsigned long long *out;
    hipMalloc(&xbuf, (size_t)wgs * 2 * HALF * 4); hipMalloc(&flags, wgs * 4); hipMalloc(&fail, 4); hipMalloc(&out, wgs * 16); hipMalloc(&sink, wgs * 4);
    std::vector<double> ex, tot, ms;
    unsigned hfail = 0;
    for (int rep = 0; rep < 12; ++rep) {
        hipMemset(flags, 0, wgs * 4); hipMemset(fail, 0, 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_pair<MODE>, dim3(wgs), dim3(NT), 0, 0, xbuf, flags, out, pxor, fail, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        std::vector<unsigned long long> h(2 * wgs);
        hipMemcpy(h.data(), out, wgs * 16, hipMemcpyDeviceToHost);
        unsigned f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost); hfail |= f;
        if (rep < 2) continue;
        double e = 0, w = 0;
        for (int i = 0; i < wgs; ++i) { e += h[2 * i]; w += h[2 * i + 1]; }
        ex.push_back(e / wgs / LAYERS * 0.01); tot.push_back(w / wgs * 0.01); ms.push_back(t * 1000);
    }
    std::sort(ex.begin(), ex.end()); std::sort(tot.begin(), tot.end()); std::sort(ms.begin(), ms.end());
    printf("%-46s pairs %3d partner wg^%d: exchange %.2f us per layer (median), %.1f us per workgroup for %d layers, launch %.1f us%s\n", what, pairs, pxor,
           ex[ex.size() / 2], tot[tot.size() / 2], LAYERS, ms[ms.size() / 2], hfail ? "  ** a pair never met **" : "");
    hipFree(xbuf); hipFree(flags); hipFree(fail); hipFree(out); hipFree(sink);
}

int main() {
    for (int pairs : {8, 48, 96, 128}) {        // 2 * pairs workgroups: multiples of 16 so that wg ^ 8 stays inside the grid
        if (pairs >= 8) run<0>("device-scope release/acquire", pairs, 8);
        run<0>("device-scope release/acquire", pairs, 1);
        if (pairs >= 8) run<1>("no data fences, non-temporal reads (bound)", pairs, 8);
    }
    return 0;
}
