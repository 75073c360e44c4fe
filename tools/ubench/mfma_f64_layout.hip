// v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane layout, probed with one-hot operands, and the Gram-matrix use of it
// (X^T X with ONE register as both operands).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_layout.hip -o tools/ubench/mfma_f64_layout && tools/ubench/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double *a, const double *b, double *out)
{
    const int lane = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = acc[r];
}
__global__ void gram(const float *x /*[rows][16]*/, int rows, double *out /*[64][4]*/)
{
    const int lane = threadIdx.x, m = lane & 15, q = lane >> 4;
    d4 acc = {0, 0, 0, 0};
    for (int s = 0; s < rows / 4; s++) {
        const double v = (double)x[(4 * s + q) * 16 + m];     // A[i = m][k = q] = X^T[m][row], B[k = q][j = m] = X[row][m]: the same value
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = acc[r];
}
int main()
{
    double *da, *db, *dout;
    hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dout, 256 * 8);
    // assumed operand layout: A[i][k] in lane 16k + i, B[k][j] in lane 16k + j.  Find where D[i][j] lands.
    int where_lane[16][16], where_reg[16][16];
    bool ok = true;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double ha[64] = {0}, hb[64] = {0}, ho[256];
            ha[16 * 1 + i] = 1.0; hb[16 * 1 + j] = 1.0;                 // k = 1
            hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
            hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
            int n = 0;
            for (int x = 0; x < 256; x++) if (ho[x] != 0.0) { where_lane[i][j] = x / 4; where_reg[i][j] = x % 4; n++; }
            if (n != 1) { ok = false; printf("(%d,%d): %d non-zeros\n", i, j, n); }
        }
    printf("operand layout A[i][k] <- lane 16k+i, B[k][j] <- lane 16k+j: %s\n", ok ? "confirmed (one result element per one-hot pair)" : "NOT confirmed");
    printf("D[i][j] lives in (lane, reg):\n");
    for (int i = 0; i < 16; i += 5) { for (int j = 0; j < 16; j += 5) printf("  D[%2d][%2d] -> lane %2d reg %d;", i, j, where_lane[i][j], where_reg[i][j]); printf("\n"); }
    // fit: lane = 16 * f(i) + j ?  reg = g(i) ?
    bool fitA = true, fitB = true;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        if (!(where_lane[i][j] == 16 * (i / 4) + j && where_reg[i][j] == i % 4)) fitA = false;      // rows 4q + r
        if (!(where_lane[i][j] == 16 * (i % 4) + j && where_reg[i][j] == i / 4)) fitB = false;      // rows 4r + q
    }
    printf("D[4q + r][m] in lane 16q + m reg r: %s;  D[4r + q][m] in lane 16q + m reg r: %s\n", fitA ? "YES" : "no", fitB ? "YES" : "no");
    const int rows = 64;
    float hx[rows * 16];
    for (int i = 0; i < rows * 16; i++) hx[i] = (float)((i * 37 % 101) - 50) / 7.0f;
    float *dx; hipMalloc(&dx, sizeof(hx)); hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gram, dim3(1), dim3(64), 0, 0, dx, rows, dout);
    double ho[256]; hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double ref = 0;
            for (int r = 0; r < rows; r++) ref += (double)hx[r * 16 + i] * (double)hx[r * 16 + j];
            worst = fmax(worst, fabs(ref - ho[where_lane[i][j] * 4 + where_reg[i][j]]));
        }
    printf("Gram matrix X^T X (64 rows x 16) with one register as both operands, read back through the probed layout: max |error| = %.3g\n", worst);
    return ok && worst < 1e-9 ? 0 : 1;
}
