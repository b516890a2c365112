// Driver of the reference-binding check (tests/test_gpu_reference_binding.py): reads two clouds, calls one of the three entry
// points of integration/mi355x_adapters.cpp the way gpumain.cpp:12-38 does, prints the result as JSON.
//   binding_check <clouds.bin> icp  <eps> <maxIterations>
//   binding_check <clouds.bin> cpd  <eps> <maxIterations> <weight> <tolerance> <approximation 0|1|2>
//   binding_check <clouds.bin> nicp <eps> <repetitions> <approximation> <subcloudSize> <seed>
// clouds.bin: int32 m, int32 n, m*3 floats (before), n*3 floats (after).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "cuda_slam_entry_points.h"

namespace Common { extern std::mt19937 mtRandom; }    // source/common/common.cpp:14

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: see the header of binding_check_main.cpp\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int m = 0, n = 0;
    if (fread(&m, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    std::vector<Common::Point_f> before(m), after(n);
    static_assert(sizeof(Common::Point_f) == 12, "Point_f is a packed xyz triple");
    if (fread(before.data(), 12, m, f) != (size_t)m || fread(after.data(), 12, n, f) != (size_t)n) return 2;
    fclose(f);
    const float eps = (float)atof(argv[3]);
    const int iters = atoi(argv[4]);
    int iterations = 0;
    float error = 0.f;
    std::pair<glm::mat3, glm::vec3> r;
    if (!strcmp(argv[2], "icp")) {
        r = GetCudaIcpTransformationMatrix(before, after, eps, iters, &iterations, &error);
    } else if (!strcmp(argv[2], "cpd") && argc >= 8) {
        r = GetCudaCpdTransformationMatrix(before, after, eps, (float)atof(argv[5]), false, iters, (float)atof(argv[6]),
                                           static_cast<Common::ApproximationType>(atoi(argv[7])), &iterations, &error, 10.0f, 8.0f);
    } else if (!strcmp(argv[2], "nicp") && argc >= 8) {
        Common::mtRandom = std::mt19937{ (unsigned)atoi(argv[7]) };          // as common.cpp:137 seeds it from "random-seed"
        r = GetCudaNicpTransformationMatrix(before, after, eps, iters, 16, static_cast<Common::ApproximationType>(atoi(argv[5])),
                                            atoi(argv[6]), &iterations, &error);
    } else {
        fprintf(stderr, "unknown method or missing arguments\n");
        return 2;
    }
    printf("\nRESULT {\"iterations\": %d, \"error\": %.9g, \"R_colmajor\": [", iterations, error);
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) printf("%s%.9g", (c || k) ? ", " : "", r.first[c][k]);
    printf("], \"t\": [%.9g, %.9g, %.9g]}\n", r.second.x, r.second.y, r.second.z);
    return 0;
}
