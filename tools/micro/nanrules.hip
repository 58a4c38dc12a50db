// nanrules.hip — what gfx950 VALU instructions return for NaN operands (sign, payload, which operand wins):
// the facts behind the NaN handling of the exact streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    float x = __uint_as_float(a[i]), y = __uint_as_float(b[i]);
    float r;
    uint32_t* o = out + i * 16;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); o[0] = __float_as_uint(r);
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); o[1] = __float_as_uint(r);
    asm volatile("v_subrev_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); o[2] = __float_as_uint(r);  // y - x
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); o[3] = __float_as_uint(r);
    asm volatile("v_med3_f32 %0, %1, -1.0, 1.0" : "=v"(r) : "v"(x)); o[4] = __float_as_uint(r);
    double dx, dy, dr;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dx) : "v"(x));
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dy) : "v"(y));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dx)); o[5] = __float_as_uint(r);
    asm volatile("v_add_f64 %0, 1.0, -%1" : "=v"(dr) : "v"(dx));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[6] = __float_as_uint(r);              // 1.0 - x
    asm volatile("v_fma_f64 %0, %1, -1.0, 1.0" : "=v"(dr) : "v"(dx));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[7] = __float_as_uint(r);              // x * -1 + 1
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(dr) : "v"(dx), "v"(dy));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[8] = __float_as_uint(r);
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(dr) : "v"(dx), "v"(dy));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[9] = __float_as_uint(r);
    asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(dr) : "v"(dx), "v"(dy));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[10] = __float_as_uint(r);             // x*y + y
    asm volatile("v_add_f32 %0, 0, %1" : "=v"(r) : "v"(x)); o[11] = __float_as_uint(r);
    asm volatile("v_sub_f32 %0, 0, %1" : "=v"(r) : "v"(x)); o[12] = __float_as_uint(r);
    asm volatile("v_mul_f32 %0, -1.0, %1" : "=v"(r) : "v"(x)); o[13] = __float_as_uint(r);
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dr) : "v"(dx), "v"(dy));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(dr)); o[14] = __float_as_uint(r);             // x - y (f64)
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x)); o[15] = __float_as_uint(r);
}
int main() {
    const uint32_t vals[] = {0x3f800000, 0x7fc00001, 0xffc00002, 0x7f800003, 0xff800004, 0x7f800000, 0xff800000, 0x00000000};
    const int nv = 8;
    uint32_t ha[64], hb[64];
    for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) { ha[i * nv + j] = vals[i]; hb[i * nv + j] = vals[j]; }
    uint32_t *da, *db, *dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, 64);
    uint32_t ho[64 * 16];
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    printf("       x        y |      add      sub   subrev      mul   med3(x)  cvt(x)   1-x(neg) fma(x,-1,1) mul64    add64  fma64(x,y,y) 0+x     0-x    -1*x   x-y(f64)\n");
    for (int i = 0; i < 64; ++i) {
        printf("%08x %08x |", ha[i], hb[i]);
        for (int q = 0; q < 15; ++q) printf(" %08x", ho[i * 16 + q]);
        printf("\n");
    }
    return 0;
}
