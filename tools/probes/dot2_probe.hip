// probe: semantics of __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* o, float* o2) {
    int i = threadIdx.x;
    o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a[i]), __builtin_bit_cast(bf2, b[i]), c[i], false);
    float alo = __uint_as_float(a[i] << 16), ahi = __uint_as_float(a[i] & 0xffff0000u);
    float blo = __uint_as_float(b[i] << 16), bhi = __uint_as_float(b[i] & 0xffff0000u);
    o2[i] = fmaf(ahi, bhi, fmaf(alo, blo, c[i]));
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
    const int N = 64;
    uint32_t ha[N], hb[N]; float hc[N], ho[N], ho2[N];
    for (int i = 0; i < N; ++i) {
        float a0 = 0.5f * i - 7, a1 = 0.25f * i + 1, b0 = 1.5f - 0.125f * i, b1 = 0.0625f * i;
        ha[i] = f2bf(a0) | ((uint32_t)f2bf(a1) << 16); hb[i] = f2bf(b0) | ((uint32_t)f2bf(b1) << 16); hc[i] = 100.f + i;
    }
    uint32_t *da, *db; float *dc, *dо, *d2;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc); hipMalloc(&dо, sizeof ho); hipMalloc(&d2, sizeof ho2);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
    k<<<1, N>>>(da, db, dc, dо, d2);
    hipMemcpy(ho, dо, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(ho2, d2, sizeof ho2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) { if (ho[i] != ho2[i]) ++bad; if (i < 6 || ho[i] != ho2[i]) printf("%d dot2=%g fma=%g\n", i, ho[i], ho2[i]); }
    printf("mismatches %d / %d\n", bad, N);
    return 0;
}
