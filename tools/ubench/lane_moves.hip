// Probe (diagnostic): which lanes the DPP controls and gfx950's row swaps used by the step kernel read.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lane_moves.hip -o tools/ubench/lane_moves && tools/ubench/lane_moves
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int x) { return __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xF, 0xF, true); }
__global__ void k(int *o)
{
    const int l = threadIdx.x;
    o[0 * 64 + l] = dpp<0x108>(l);
    o[1 * 64 + l] = dpp<0x138>(l);
    o[2 * 64 + l] = dpp<0x141>(l);
    o[3 * 64 + l] = dpp<0xB1>(l);
    o[4 * 64 + l] = dpp<0x4E>(l);
    o[5 * 64 + l] = (int)__builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)l, false, false)[1];
    o[6 * 64 + l] = (int)__builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)l, false, false)[1];
    o[7 * 64 + l] = (int)__builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)l, false, false)[0];
}
int main()
{
    int *d, h[8 * 64];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[8] = {"row_shl:8", "wave_shr:1", "row_half_mirror", "quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "permlane16_swap[1]", "permlane32_swap[1]", "permlane16_swap[0]"};
    for (int r = 0; r < 8; r++) { printf("%-20s", nm[r]); for (int l = 0; l < 64; l++) printf(" %d", h[r * 64 + l]); printf("\n"); }
    return 0;
}
