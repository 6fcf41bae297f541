// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the tree kernels' access shapes (the microarchitecture
// guide calibrates them for wide streaming accesses only and asks for this before absolutes are trusted):
//   k_rec128   every group of 8 lanes reads one whole 128-byte record (one 16-byte load per lane), records in scrambled order
//   k_sect16   lane 0 of every group of 8 reads 16 bytes of a 128-byte record (the 8-byte read-modify-writes of a backup and the
//              small per-slot rows touch a line like this)
//   k_rmw8     one lane per record reads 8 bytes, adds, writes 8 bytes back
//   k_stream   plain coalesced 16-byte-per-lane streaming read (the guide's reference shape: FETCH_SIZE = bytes / 2)
// Each kernel touches every record exactly once (R records, far more than the caches hold), so the true byte counts are known.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- ./fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ unsigned scramble(unsigned i, unsigned n) {      // a bijection on [0, n), n a power of two
    i = (i * 2654435761u) & (n - 1);
    return i ^ (i >> 7);
}
__global__ void k_rec128(const int4 *rec, unsigned n, int *sink) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, grp = t >> 3, lane = t & 7;
    if (grp >= n) return;
    const int4 v = rec[(size_t)scramble(grp, n) * 8 + lane];
    if (v.x == 0x7fffffff) sink[0] = v.y + v.z + v.w;
}
__global__ void k_sect16(const int4 *rec, unsigned n, int *sink) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, grp = t >> 3, lane = t & 7;
    if (grp >= n || lane != 0) return;
    const int4 v = rec[(size_t)scramble(grp, n) * 8 + 3];
    if (v.x == 0x7fffffff) sink[0] = v.y + v.z + v.w;
}
__global__ void k_rmw8(int2 *rec, unsigned n) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, grp = t >> 3, lane = t & 7;
    if (grp >= n || lane != 0) return;
    int2 *p = rec + (size_t)scramble(grp, n) * 16 + 5;
    int2 v = *p;
    v.x += 1;
    *p = v;
}
__global__ void k_stream(const int4 *rec, size_t n16, int *sink) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n16) return;
    const int4 v = rec[t];
    if (v.x == 0x7fffffff) sink[0] = v.y + v.z + v.w;
}
int main() {
    const unsigned n = 1u << 22;                       // 4 Mi records x 128 B = 512 MiB
    int4 *rec; int *sink;
    hipMalloc(&rec, (size_t)n * 128); hipMemset(rec, 0, (size_t)n * 128); hipMalloc(&sink, 64);
    hipDeviceSynchronize();
    const unsigned threads = n * 8;
    k_rec128<<<threads / 256, 256>>>(rec, n, sink);
    k_sect16<<<threads / 256, 256>>>(rec, n, sink);
    k_rmw8<<<threads / 256, 256>>>(reinterpret_cast<int2 *>(rec), n);
    k_stream<<<threads / 256, 256>>>(rec, (size_t)n * 8, sink);
    hipDeviceSynchronize();
    printf("records %u: k_rec128 reads %.1f MiB; k_sect16 reads %.1f MiB of payload (%.1f MiB of 64-byte sectors); "
           "k_rmw8 reads and writes %.1f MiB (%.1f MiB of 64-byte sectors each way); k_stream reads %.1f MiB\n",
           n, n * 128.0 / 1048576, n * 16.0 / 1048576, n * 64.0 / 1048576, n * 8.0 / 1048576, n * 64.0 / 1048576, n * 128.0 / 1048576);
    return 0;
}
