// What does ds_read_b64_tr_b16 return?  (gfx950; tools only.)  LDS holds u16 value = element index; lane l supplies the byte
// address of elements [4 l, 4 l + 4) -- or, in pattern 1, of a [32 rows][pitch] tile the way a transposed MFMA operand wants it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/trp tools/tr_b16_probe.hip && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(int pattern, int pitch_el, uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr_el;
    if (pattern == 0) addr_el = 4 * l;
    else {
        // lane p of a 16-lane group: row (p >> 2) + 4 * group, column chunk (p & 3) -> 4 rows x 16 columns per group
        const int g = l >> 4, p = l & 15;
        addr_el = ((p >> 2) + 4 * g) * pitch_el + (p & 3) * 4;
    }
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t *)lds + addr_el * 2;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int pattern = 0; pattern < 2; pattern++) {
        const int pitch = 64;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pattern, pitch, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d (pitch %d elements): lane: 4 returned element indices\n", pattern, pitch);
        for (int l = 0; l < 64; l++) {
            printf("  %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
