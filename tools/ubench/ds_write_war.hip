// ds_write_war.hip — does gfx950 read the DATA registers of an LDS store after the instruction has issued, without interlocking a
// following VALU write to them?  (Round-4 bisection of the k_gat3x function-form miscomputation: the failing schedule overwrote
// v[10:11] in the very next VALU instruction after `ds_write_b128 v126, v[10:13]`; -amdgpu-waitcnt-forcezero made it right.)
// Each wave stores a known 16/8/4-byte pattern from fixed registers, overwrites those registers with junk after `gap` filler
// instructions, reads the LDS back and counts lanes whose stored value is not the pattern.  8 waves per workgroup (2 per SIMD), one
// workgroup per CU, all hammering the LDS.   hipcc --offload-arch=gfx950 -O3 -o ds_write_war ds_write_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int WIDTH, int GAP, int FILL>      // FILL 0: s_nop gap, 1: independent VALU gap
__global__ __launch_bounds__(512) void k(unsigned *bad, int iters)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const unsigned addr = tid * 16;                      // private 16 bytes per thread
    unsigned nbad = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned a = 0x10000000u + it * 4 + tid, b = a ^ 0x5a5a5a5au, c = a + 0x01010101u, d = ~a;
        const unsigned junk = 0xdead0000u + it;
        unsigned filler = a;
        // pattern into v[20:23], the store, GAP fillers, then junk over the data registers
        asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 7\n\ts_nop 7" :: "v"(addr), "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23");
        if (WIDTH == 16) asm volatile("ds_write_b128 %0, v[20:23]" :: "v"(addr) : "memory", "v20", "v21", "v22", "v23");
        else if (WIDTH == 8) asm volatile("ds_write_b64 %0, v[20:21]" :: "v"(addr) : "memory", "v20", "v21");
        else asm volatile("ds_write_b32 %0, v20" :: "v"(addr) : "memory", "v20");
#pragma unroll
        for (int g = 0; g < GAP; g++) {
            if (FILL) asm volatile("v_add_u32 %0, %0, %0" : "+v"(filler));
            else asm volatile("s_nop 0");
        }
        asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\ts_waitcnt lgkmcnt(0)" :: "v"(junk) : "v20", "v21", "v22", "v23", "memory");
        const volatile unsigned *rp = reinterpret_cast<const volatile unsigned *>(smem + addr);
        const unsigned rx = rp[0], ry = rp[1], rz = rp[2], rw = rp[3];
        bool ok = rx == a;
        if (WIDTH >= 8) ok = ok && ry == b;
        if (WIDTH == 16) ok = ok && rz == c && rw == d;
        nbad += !ok;
        if (filler == 0x12345u) nbad += 1000;            // keeps the fillers
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int WIDTH, int GAP, int FILL>
static int run(unsigned *d_bad, int iters)
{
    CHK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL((k<WIDTH, GAP, FILL>), dim3(256), dim3(512), 512 * 16, 0, d_bad, iters);
    CHK(hipDeviceSynchronize());
    unsigned h = 0;
    CHK(hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost));
    printf("ds_write_b%-3d  gap %d %s : %u corrupted stores of %lld\n", WIDTH * 8, GAP, FILL ? "VALU " : "s_nop", h, (long long)iters * 256 * 512);
    return 0;
}

int main()
{
    unsigned *d_bad;
    CHK(hipMalloc(&d_bad, 4));
    const int iters = 20000;
    run<16, 0, 0>(d_bad, iters); run<16, 1, 0>(d_bad, iters); run<16, 2, 0>(d_bad, iters); run<16, 4, 0>(d_bad, iters); run<16, 8, 0>(d_bad, iters);
    run<16, 1, 1>(d_bad, iters); run<16, 2, 1>(d_bad, iters); run<16, 3, 1>(d_bad, iters); run<16, 4, 1>(d_bad, iters);
    run<8, 0, 0>(d_bad, iters); run<8, 1, 0>(d_bad, iters); run<8, 2, 0>(d_bad, iters); run<8, 1, 1>(d_bad, iters);
    run<4, 0, 0>(d_bad, iters); run<4, 1, 0>(d_bad, iters);
    return 0;
}
