// The reference-side binding of INTEGRATION.md section 1, as one compilable file: the bodies a maintainer of Sliwson/cuda-slam
// puts behind GetCudaIcpTransformationMatrix / GetCudaCpdTransformationMatrix / GetCudaNicpTransformationMatrix to run them on
// libmislam.so.  Written against the reference's own headers (common.h, glm); oracle/Makefile compiles it with them and links
// the reference's common.cpp object, and tests/test_gpu_reference_binding.py runs the result on the GPU box.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <glm/gtc/type_ptr.hpp>

#include "cuda_slam_entry_points.h"   // in the reference tree: icpcuda.cuh, cpdcuda.cuh, nicpcuda.cuh
#include "mi_slam.h"

static mi_ctx* Ctx()
{
    static mi_ctx* ctx = [] {
        mi_ctx* c = nullptr;
        if (mi_ctx_create(/*device*/ 0, &c) != MI_OK) {           // same policy as checkCudaErrors (helper_cuda.h:567-573)
            fprintf(stderr, "MI355X error: %s\n", mi_last_error());
            exit(EXIT_FAILURE);
        }
        return c;
    }();
    return ctx;
}

std::pair<glm::mat3, glm::vec3> GetCudaIcpTransformationMatrix(
    const std::vector<Common::Point_f>& cloudBefore, const std::vector<Common::Point_f>& cloudAfter,
    float eps, int maxIterations, int* iterations, float* error)
{
    mi_icp_params p;
    mi_icp_params_cuda_slam(&p);      // the GPU reference's driver rules (icpcuda.cu:8-58); mi_icp_params_default = cpu-slam's
    p.eps = eps;
    p.max_iterations = maxIterations; // -1 = unbounded, as gpumain.cpp:14 passes it
    p.verbose = 1;                    // "Iteration: %d, error: %f" progress lines
    glm::mat4 T;
    if (mi_icp_register(Ctx(), reinterpret_cast<const float*>(cloudBefore.data()), (int)cloudBefore.size(),
                        reinterpret_cast<const float*>(cloudAfter.data()), (int)cloudAfter.size(),
                        &p, glm::value_ptr(T), iterations, error) != MI_OK) {
        fprintf(stderr, "MI355X error: %s\n", mi_last_error());
        exit(EXIT_FAILURE);
    }
    return Common::ConvertToRotationTranslationPair(T);            // common.cpp:360-365
}

std::pair<glm::mat3, glm::vec3> GetCudaCpdTransformationMatrix(
    const std::vector<Common::Point_f>& cloudBefore, const std::vector<Common::Point_f>& cloudAfter,
    float eps, float weight, bool const_scale, int maxIterations, float tolerance, Common::ApproximationType fgt,
    int* iterations, float* error, const float& ratioOfFarField, const float& orderOfTruncation)
{
    mi_cpd_params p;
    mi_cpd_params_default(&p);
    p.eps = eps; p.weight = weight; p.const_scale = const_scale; p.max_iterations = maxIterations; p.tolerance = tolerance;
    p.approximation = (int)fgt;                       // enumerators.h:18-23 and MI_CPD_APPROX_* share the numbering
    p.fgt_ratio_of_far_field = ratioOfFarField;       // the FGT E-step runs on the device too (the reference runs it on the CPU)
    p.fgt_order_of_truncation = (int)orderOfTruncation;
    // The initial sigma^2.  cpu-slam's is a sequential fp32 running sum that saturates (3.604 instead of 12.943 on bunny,
    // coherentpointdrift.cpp:126-139) and its whole trajectory starts there; cuda-slam's thrust reduction (cpdcuda.cu:65-78) gives
    // roughly the exact value.  The parity target of this build is cpu-slam, so that is the default here (the device computes that
    // very sum in ~3 ms for bunny-sized clouds, ~15 us per million pairs); MI355X_CPD_SIGMA=exact selects what cpdcuda.cu computes.
    {
        const char* mode = getenv("MI355X_CPD_SIGMA");
        const double pairs = (double)cloudBefore.size() * (double)cloudAfter.size();
        const bool cpu = mode ? (mode[0] == 'c') : pairs <= 4e9;
        p.sigma2_mode = cpu ? MI_SIGMA2_CPU_SEQUENTIAL : MI_SIGMA2_EXACT;
    }
    glm::mat4 sRt;
    if (mi_cpd_register(Ctx(), reinterpret_cast<const float*>(cloudBefore.data()), (int)cloudBefore.size(),
                        reinterpret_cast<const float*>(cloudAfter.data()), (int)cloudAfter.size(),
                        &p, glm::value_ptr(sRt), nullptr, iterations, error) != MI_OK) {
        fprintf(stderr, "MI355X error: %s\n", mi_last_error());
        exit(EXIT_FAILURE);
    }
    return Common::ConvertToRotationTranslationPair(sRt);          // mat3 slot holds scale*R, as cpdcuda.cu:360 returns
}

std::pair<glm::mat3, glm::vec3> GetCudaNicpTransformationMatrix(
    const std::vector<Common::Point_f>& before, const std::vector<Common::Point_f>& after, float eps, int maxRepetitions,
    int /*batchSize*/, Common::ApproximationType approximationType, const int subcloudSize, int* repetitions, float* error)
{
    mi_nicp_params p;
    mi_nicp_params_default(&p);
    p.eps = eps; p.max_repetitions = maxRepetitions; p.approximation = (int)approximationType;
    // the reference's draws, in its order (noniterative.cpp:213-222): subcloud first (none if it is the whole cloud), then one
    // permutation per repetition, of which only the first three entries matter
    std::vector<int> subcloud;
    if (subcloudSize < (int)before.size()) { subcloud = Common::GetRandomPermutationVector((int)before.size()); subcloud.resize(subcloudSize); }
    const int reps = maxRepetitions == -1 ? 20 : maxRepetitions, size = (int)std::min(before.size(), after.size());
    std::vector<int> heads(3 * reps);
    for (int r = 0; r < reps; r++) { auto perm = Common::GetRandomPermutationVector(size); std::copy_n(perm.begin(), 3, &heads[3 * r]); }
    glm::mat4 T;
    if (mi_nicp_register(Ctx(), reinterpret_cast<const float*>(before.data()), (int)before.size(),
                         reinterpret_cast<const float*>(after.data()), (int)after.size(), &p, heads.data(),
                         subcloud.empty() ? nullptr : subcloud.data(), subcloud.empty() ? (int)before.size() : (int)subcloud.size(),
                         glm::value_ptr(T), repetitions, error) != MI_OK) {
        fprintf(stderr, "MI355X error: %s\n", mi_last_error());
        exit(EXIT_FAILURE);
    }
    return Common::ConvertToRotationTranslationPair(T);
}
