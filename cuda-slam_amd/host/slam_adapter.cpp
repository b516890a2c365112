#include "slam_adapter.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "../../include/mi_slam.h"
#include "cloud_io.h"

using namespace Common;

namespace {

SlamRules g_rules = SlamRules::CudaSlam;
float g_max_distance_squared = 1000.f;
int g_device = 0;
mi_ctx* g_ctx = nullptr;

// checkCudaErrors behaviour (include/helper_cuda.h:567-573): report and leave
void check(int rc, const char* what)
{
    if (rc == MI_OK) return;
    fprintf(stderr, "MI355X device error at %s: code=%d \"%s\"\n", what, rc, mi_last_error());
    exit(EXIT_FAILURE);
}

mi_ctx* context()
{
    // created once per process (the reference creates and destroys its cuBLAS/cuSOLVER handles and every buffer per call)
    if (!g_ctx) {
        check(mi_ctx_create(g_device, &g_ctx), "mi_ctx_create");
        atexit([] { mi_ctx_destroy(g_ctx); g_ctx = nullptr; });
    }
    return g_ctx;
}

std::pair<Mat3, Vec3> split(const float T[16])   // ConvertToRotationTranslationPair, common.cpp:360-365
{
    Mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R[c][r] = T[4 * c + r];
    return {R, Vec3{T[12], T[13], T[14]}};
}

}  // namespace

void SetSlamRules(SlamRules rules, float max_distance_squared)
{
    g_rules = rules;
    g_max_distance_squared = max_distance_squared;
}

void SetSlamDevice(int device) { g_device = device; }
mi_ctx* GetSlamContext() { return context(); }

std::pair<Mat3, Vec3> GetCudaIcpTransformationMatrix(const std::vector<Point_f>& cloudBefore, const std::vector<Point_f>& cloudAfter,
                                                    float eps, int maxIterations, int* iterations, float* error)
{
    mi_icp_params p;
    if (g_rules == SlamRules::CudaSlam) mi_icp_params_cuda_slam(&p);
    else mi_icp_params_default(&p);
    p.eps = eps;
    p.max_iterations = maxIterations;
    p.max_distance_squared = g_max_distance_squared;
    p.verbose = 1;   // the reference prints one line per iteration (icpcuda.cu:39)
    float T[16];
    // Point_f is the 12-byte xyz triple the C ABI expects: the vectors are passed as they are
    check(mi_icp_register(context(), reinterpret_cast<const float*>(cloudBefore.data()), (int)cloudBefore.size(),
                          reinterpret_cast<const float*>(cloudAfter.data()), (int)cloudAfter.size(), &p, T, iterations, error),
          "mi_icp_register");
    return split(T);
}

std::pair<Mat3, Vec3> GetCudaCpdTransformationMatrix(const std::vector<Point_f>& cloudBefore, const std::vector<Point_f>& cloudAfter,
                                                    float eps, float weight, bool const_scale, int maxIterations, float tolerance,
                                                    ApproximationType fgt, int* iterations, float* error, const float& ratioOfFarField,
                                                    const float& orderOfTruncation)
{
    mi_cpd_params p;
    mi_cpd_params_default(&p);
    // enumerators.h:18-23 and MI_CPD_APPROX_* share the numbering (none, full, hybrid)
    p.approximation = fgt == ApproximationType::Full ? MI_CPD_APPROX_FULL : fgt == ApproximationType::Hybrid ? MI_CPD_APPROX_HYBRID : MI_CPD_APPROX_NONE;
    p.fgt_ratio_of_far_field = ratioOfFarField;
    p.fgt_order_of_truncation = (int)orderOfTruncation;      // cpdutils.cpp:37 narrows the float the same way
    p.eps = eps;
    p.weight = weight;
    p.const_scale = const_scale ? 1 : 0;
    p.max_iterations = maxIterations;
    p.tolerance = tolerance;
    p.verbose = 1;   // cpdcuda.cu:357
    // the initial sigma^2 as cpu-slam (the parity target) computes it -- a saturating sequential fp32 sum,
    // coherentpointdrift.cpp:126-139 -- which the device computes in ~15 us per million pairs; MI355X_CPD_SIGMA=exact selects what
    // cpdcuda.cu computes (same policy as integration/mi355x_adapters.cpp)
    {
        const char* mode = getenv("MI355X_CPD_SIGMA");
        const double pairs = (double)cloudBefore.size() * (double)cloudAfter.size();
        p.sigma2_mode = (mode ? mode[0] == 'c' : pairs <= 4e9) ? MI_SIGMA2_CPU_SEQUENTIAL : MI_SIGMA2_EXACT;
    }
    float T[16];
    check(mi_cpd_register(context(), reinterpret_cast<const float*>(cloudBefore.data()), (int)cloudBefore.size(),
                          reinterpret_cast<const float*>(cloudAfter.data()), (int)cloudAfter.size(), &p, T, nullptr, iterations, error),
          "mi_cpd_register");
    return split(T);
}

std::pair<Mat3, Vec3> GetCudaNicpTransformationMatrix(const std::vector<Point_f>& before, const std::vector<Point_f>& after, float eps,
                                                     int maxRepetitions, int batchSize, ApproximationType approximationType,
                                                     const int subcloudSize, int* repetitions, float* error)
{
    (void)batchSize;
    mi_nicp_params p;
    mi_nicp_params_default(&p);
    p.eps = eps;
    p.max_repetitions = maxRepetitions;
    p.approximation = approximationType == ApproximationType::Full ? MI_CPD_APPROX_FULL
                    : approximationType == ApproximationType::Hybrid ? MI_CPD_APPROX_HYBRID : MI_CPD_APPROX_NONE;
    p.verbose = 1;
    // The random draws, in the reference's order (noniterative.cpp:213-222): first the comparison subcloud -- a permutation of
    // |before| cut to subcloudSize, or nothing at all when the subcloud is the whole cloud (common.cpp:25-37) -- then one
    // permutation of min(|before|, |after|) per repetition, of which the registration needs the first three entries only.
    std::vector<int> subcloud;
    if (subcloudSize < (int)before.size()) {
        subcloud = GetRandomPermutationVector((int)before.size());
        subcloud.resize((size_t)std::max(subcloudSize, 0));
    }
    const int reps = maxRepetitions == -1 ? 20 : std::max(maxRepetitions, 0);
    const int size = (int)std::min(before.size(), after.size());
    std::vector<int> heads(3 * (size_t)reps);
    for (int r = 0; r < reps; r++) {
        const std::vector<int> perm = GetRandomPermutationVector(size);
        for (int k = 0; k < 3 && k < size; k++) heads[3 * (size_t)r + k] = perm[(size_t)k];
    }
    float T[16];
    check(mi_nicp_register(context(), reinterpret_cast<const float*>(before.data()), (int)before.size(),
                           reinterpret_cast<const float*>(after.data()), (int)after.size(), &p, heads.data(),
                           subcloud.empty() ? nullptr : subcloud.data(), subcloud.empty() ? (int)before.size() : (int)subcloud.size(), T,
                           repetitions, error),
          "mi_nicp_register");
    return split(T);
}

std::pair<Mat3, Vec3> GetGpuSlamResult(const CpuCloud& before, const CpuCloud& after, Configuration configuration, int* iterations,
                                      float* error)
{
    const int maxIterations = configuration.MaxIterations.has_value() ? configuration.MaxIterations.value() : -1;   // gpumain.cpp:14
    switch (configuration.ComputationMethod_) {
    case ComputationMethod::Cpd:
        return GetCudaCpdTransformationMatrix(before, after, configuration.ConvergenceEpsilon, configuration.CpdWeight,
                                              configuration.CpdConstScale, maxIterations, configuration.CpdTolerance,
                                              configuration.ApproximationType_, iterations, error, configuration.RatioOfFarField,
                                              (float)configuration.OrderOfTruncation);
    case ComputationMethod::NoniterativeIcp:
        return GetCudaNicpTransformationMatrix(before, after, configuration.ConvergenceEpsilon, configuration.NicpIterations,
                                               configuration.NicpBatchSize, configuration.ApproximationType_,
                                               configuration.NicpSubcloudSize, iterations, error);
    case ComputationMethod::Icp:
    default:
        return GetCudaIcpTransformationMatrix(before, after, configuration.ConvergenceEpsilon, maxIterations, iterations, error);
    }
}
