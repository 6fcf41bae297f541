// Layout probe for v_mfma_f32_4x4x1_16B_f32 on gfx950: D = A(4x1) * B(1x4) per block, 16 blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}
int main() {
    float ha[64], hb[64], hd[256], he[256];
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + i; hb[i] = 1; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) { ha[i] = 1; hb[i] = 1 + i; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(he, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int v = 0; v < 4; ++v) printf("  v%d=a[%2d]*b[%2d]", v, (int)hd[l * 4 + v] - 1, (int)he[l * 4 + v] - 1);
        printf("\n");
    }
    return 0;
}
