// What v_cvt_pk_bf8_f32 writes with / without the destination op_sel bit on gfx950 (probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* o) {
    float a = 1.0f, b = 2.0f, c = 4.0f, d = 8.0f;  // e5m2: 0x3c 0x40 0x44 0x48
    uint32_t r0, r1, r2, r3, r4;
    asm volatile(
        "v_mov_b32 %0, 0xaaaaaaaa\n\tv_cvt_pk_bf8_f32 %0, %5, %6\n\t"
        "v_mov_b32 %1, 0xaaaaaaaa\n\tv_cvt_pk_bf8_f32 %1, %7, %8 op_sel:[0,0,1]\n\t"
        "v_mov_b32 %2, 0xaaaaaaaa\n\tv_cvt_pk_bf8_f32 %2, %5, %6\n\tv_cvt_pk_bf8_f32 %2, %7, %8 op_sel:[0,0,1]\n\t"
        "v_mov_b32 %3, 0xaaaaaaaa\n\tv_cvt_pk_bf8_f32 %3, %7, %8 op_sel:[0,0,1]\n\tv_cvt_pk_bf8_f32 %3, %5, %6\n\t"
        "v_mov_b32 %4, 0xaaaaaaaa\n\tv_cvt_pk_bf8_f32 %4, %5, %6\n\ts_nop 4\n\tv_cvt_pk_bf8_f32 %4, %7, %8 op_sel:[0,0,1]\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a), "v"(b), "v"(c), "v"(d));
    if (threadIdx.x == 0) { o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; }
}
int main() {
    uint32_t* d; hipMalloc((void**)&d, 64); k<<<1, 64>>>(d); uint32_t h[5]; hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("lo only %08x | hi only %08x | lo,hi %08x | hi,lo %08x | lo,nop,hi %08x   (want ....403c | 4844.... | 4844403c)\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
