// developer probe: semantics of the DPP whole-wave shifts on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *a, int *b, int *c, int *d, int *e)
{
    int v = threadIdx.x + 100;
    a[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    b[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    c[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);  // wave_ror:1
    d[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
    e[threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);   // shr, old = self
}
int main()
{
    int *p; hipMalloc(&p, 5 * 64 * 4);
    k<<<1, 64>>>(p, p + 64, p + 128, p + 192, p + 256);
    int h[320]; hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost);
    const char *n[5] = {"wave_shr:1", "wave_shl:1", "wave_ror:1", "wave_rol:1", "shr old=self"};
    for (int r = 0; r < 5; r++) {
        printf("%s:", n[r]);
        for (int i = 0; i < 64; i++) if (i < 3 || (i > 13 && i < 19) || (i > 29 && i < 35) || i > 60) printf(" [%d]=%d", i, h[r * 64 + i]);
        printf("\n");
    }
    return 0;
}
