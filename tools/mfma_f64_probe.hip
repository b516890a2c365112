// Which lane holds which element of v_mfma_f64_4x4x4_4b_f64 (round 6: the moments row of an ICP iteration as four such products, icp_rows.hpp).
// One product per (la, lb): A = 1 in lane la only, B = 1 in lane lb only; the lane(s) of D that read 1 say which (block, i, k) / (block, k, j)
// the two operand lanes are.   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out)          // out[la * 64 + lb] = lane of D that is non-zero (or -1; -2: several)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m == 0ull ? -1 : (__builtin_popcountll(m) == 1 ? __builtin_ctzll(m) : -2);
        }
}
int main()
{
    int* d; static int h[4096];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // guess: A lane = 16 b + 4 k + i, B lane = 16 b + 4 k + j, D lane = 16 b + 4 i + j  -- checked against every pair
    int bad_guess1 = 0, bad_guess2 = 0;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const int ba = la / 16, ka = (la / 4) % 4, ia = la % 4, bb = lb / 16, kb = (lb / 4) % 4, jb = lb % 4;
            const int e1 = (ba == bb && ka == kb) ? 16 * ba + 4 * ia + jb : -1;      // D[i][j] at lane 4 i + j
            const int e2 = (ba == bb && ka == kb) ? 16 * ba + 4 * jb + ia : -1;      // D[i][j] at lane 4 j + i
            bad_guess1 += h[la * 64 + lb] != e1; bad_guess2 += h[la * 64 + lb] != e2;
        }
    printf("layout A=16b+4k+i, B=16b+4k+j: D lane 16b+4i+j mismatches %d; D lane 16b+4j+i mismatches %d\n", bad_guess1, bad_guess2);
    for (int la = 0; la < 8; la++) { printf("la %d:", la); for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] != -1) printf(" (lb %d -> D %d)", lb, h[la * 64 + lb]); printf("\n"); }
    for (int la = 16; la < 20; la++) { printf("la %d:", la); for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] != -1) printf(" (lb %d -> D %d)", lb, h[la * 64 + lb]); printf("\n"); }
    return 0;
}
