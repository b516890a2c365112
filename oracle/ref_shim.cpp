// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product path.
//
// extern "C" entry points over the reference's own CPU implementation (source/cpu-slam + source/common),
// compiled by oracle/Makefile from the sources where they lie under /root/reference into
// oracle/_ref/libref_cpuslam.so.  Nothing here re-implements the algorithm: every function forwards to the
// reference function named in its comment.  The only code of our own is the float-array <-> std::vector<Point_f>
// marshalling and the clouds-from-config sequence, which calls the reference's own NormalizeCloud / std::shuffle on
// Common::mtRandom / GetTransformedCloud in the order source/common/common.cpp:134-190 does (the assimp loader the
// reference's LoadCloud needs ships only as a Windows .lib, so the raw points come in from our OBJ reader instead).
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>

#include "common.h"
#include "testutils.h"
#include "basicicp.h"
#include "coherentpointdrift.h"
#include "cpdutils.h"
#include "fgt.h"
#include "fgt_model.h"
#include "noniterative.h"
#include "nicputils.h"

using Common::Point_f;

namespace Common {
extern std::mt19937 mtRandom;      // source/common/common.cpp:14
extern unsigned int randomSeed;    // source/common/common.cpp:13
}

namespace CoherentPointDrift {
// external-linkage helpers declared at the top of source/cpu-slam/coherentpointdrift.cpp:14-42
float CalculateSigmaSquared(const std::vector<Point_f>& cloudBefore, const std::vector<Point_f>& cloudAfter);
Probabilities ComputePMatrix(const std::vector<Point_f>& cloudTransformed, const std::vector<Point_f>& cloudAfter,
                             const float& constant, const float& sigmaSquared, const bool& doTruncate, float truncate);
void MStep(const std::vector<Point_f>& cloudBefore, const std::vector<Point_f>& cloudAfter,
           const Probabilities& probabilities, bool const_scale, glm::mat3* rotationMatrix,
           glm::vec3* translationVector, float* scale, float* sigmaSquared);
}

namespace {
std::vector<Point_f> to_cloud(const float* xyz, int n)
{
    std::vector<Point_f> c(n);
    for (int i = 0; i < n; i++) c[i] = Point_f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return c;
}
void from_cloud(const std::vector<Point_f>& c, float* xyz)
{
    for (size_t i = 0; i < c.size(); i++) { xyz[3 * i] = c[i].x; xyz[3 * i + 1] = c[i].y; xyz[3 * i + 2] = c[i].z; }
}
// glm::mat3 is column-major m[col][row]; out9 is column-major too (out9[3*col+row]).
void from_mat3(const glm::mat3& m, float* out9)
{
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) out9[3 * c + r] = m[c][r];
}
glm::mat3 to_mat3(const float* in9)
{
    glm::mat3 m;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m[c][r] = in9[3 * c + r];
    return m;
}
}

#define REF_API extern "C" __attribute__((visibility("default")))

// Clouds-from-config for the "same file, explicit transformation" case: common.cpp:134-190.
// rot9_rowmajor is the JSON "rotation" array as written in the config file (configparser.cpp:139-141 stores
// rotationMatrix[y][x] = rotation[x*3+y], i.e. the JSON is row-major), scale multiplies it (configparser.cpp:147).
REF_API void ref_clouds_from_config(const float* raw_xyz, int n, int has_spread, float spread, unsigned seed,
                                    const float* rot9_rowmajor, const float* trans3, float scale,
                                    float* before_xyz, float* after_xyz)
{
    Common::randomSeed = seed;                                   // common.cpp:136
    Common::mtRandom = std::mt19937{ seed };                     // common.cpp:137
    auto before = to_cloud(raw_xyz, n);
    auto after = before;                                         // common.cpp:142 (sameClouds)
    if (has_spread) {                                            // common.cpp:158-163
        before = Common::NormalizeCloud(before, spread);
        after = Common::NormalizeCloud(after, spread);
    }
    std::shuffle(before.begin(), before.end(), Common::mtRandom); // common.cpp:166
    std::shuffle(after.begin(), after.end(), Common::mtRandom);   // common.cpp:167
    glm::mat3 R;
    for (int x = 0; x < 3; x++) for (int y = 0; y < 3; y++) R[y][x] = rot9_rowmajor[x * 3 + y];
    R = scale * R;
    glm::vec3 t(trans3[0], trans3[1], trans3[2]);
    after = Common::GetTransformedCloud(after, R, t);            // common.cpp:187-190
    from_cloud(before, before_xyz);
    from_cloud(after, after_xyz);
}

// The whole of GetCloudsFromConfig after LoadCloud (common.cpp:139-190) for in-memory raw clouds (LoadCloud itself needs assimp,
// Windows .lib only): every stage is the reference's own function, in the reference's order, on the reference's generators
// (mtRandom for the permutations, rand() -- seeded like mainwrapper.cpp:17-18 -- for the noise and outlier draws).
// resize_* < 0 / noise_share_* < 0: option absent.  Outputs hold (resized size + outliers) points; the counts are returned.
REF_API void ref_clouds_from_config_full(const float* raw_before, int nb_raw, const float* raw_after, int na_raw, int resize_before,
                                         int resize_after, int has_spread, float spread, unsigned seed, float noise_share_before,
                                         float noise_intensity_before, float noise_share_after, float noise_intensity_after,
                                         int outliers_before, int outliers_after, const float* rot9_colmajor, const float* trans3,
                                         float* before_xyz, int* nb_out, float* after_xyz, int* na_out)
{
    srand(seed);                                                 // mainwrapper.cpp:17-18
    Common::randomSeed = seed;                                   // common.cpp:136
    Common::mtRandom = std::mt19937{ seed };                     // common.cpp:137
    auto before = to_cloud(raw_before, nb_raw);
    auto after = raw_after ? to_cloud(raw_after, na_raw) : before;   // common.cpp:141-142
    if (resize_before >= 0) before = Common::GetSubcloud(before, resize_before);   // common.cpp:145-149
    if (resize_after >= 0) after = Common::GetSubcloud(after, resize_after);       // common.cpp:151-155
    if (has_spread) {                                            // common.cpp:158-163
        before = Common::NormalizeCloud(before, spread);
        after = Common::NormalizeCloud(after, spread);
    }
    std::shuffle(before.begin(), before.end(), Common::mtRandom); // common.cpp:166
    std::shuffle(after.begin(), after.end(), Common::mtRandom);   // common.cpp:167
    if (noise_share_before >= 0.f) before = Common::AddNoiseToCloud(before, noise_share_before, noise_intensity_before);   // :170-173
    if (noise_share_after >= 0.f) after = Common::AddNoiseToCloud(after, noise_share_after, noise_intensity_after);        // :175-178
    before = Common::AddOutliersToCloud(before, outliers_before);   // common.cpp:180
    after = Common::AddOutliersToCloud(after, outliers_after);      // common.cpp:181
    after = Common::GetTransformedCloud(after, to_mat3(rot9_colmajor), glm::vec3(trans3[0], trans3[1], trans3[2]));   // :184-190
    from_cloud(before, before_xyz);
    from_cloud(after, after_xyz);
    *nb_out = (int)before.size();
    *na_out = (int)after.size();
}

// The same stage with the RANDOM known transformation of the reference's benchmark sets (configuration.TransformationParameters =
// (rotation range, translation range): testset.cpp:62, :102, :146-176): Tests::GetRandomRotationMatrix / GetRandomTranslationVector
// (testutils.cpp:43-55) drawn from rand() where GetCloudsFromConfig draws them -- behind the noise and outlier draws (common.cpp:193-204).
// The drawn rotation (column-major) and translation are returned too.
REF_API void ref_clouds_from_config_random(const float* raw_before, int nb_raw, const float* raw_after, int na_raw, int resize_before,
                                           int resize_after, int has_spread, float spread, unsigned seed, float rot_range, float trans_range,
                                           float* before_xyz, int* nb_out, float* after_xyz, int* na_out, float* rot9_colmajor, float* trans3)
{
    srand(seed);                                                 // mainwrapper.cpp:17-18
    Common::randomSeed = seed;                                   // common.cpp:136
    Common::mtRandom = std::mt19937{ seed };                     // common.cpp:137
    auto before = to_cloud(raw_before, nb_raw);
    auto after = raw_after ? to_cloud(raw_after, na_raw) : before;   // common.cpp:141-142
    if (resize_before >= 0) before = Common::GetSubcloud(before, resize_before);   // common.cpp:145-149
    if (resize_after >= 0) after = Common::GetSubcloud(after, resize_after);       // common.cpp:151-155
    if (has_spread) {                                            // common.cpp:158-163
        before = Common::NormalizeCloud(before, spread);
        after = Common::NormalizeCloud(after, spread);
    }
    std::shuffle(before.begin(), before.end(), Common::mtRandom); // common.cpp:166
    std::shuffle(after.begin(), after.end(), Common::mtRandom);   // common.cpp:167
    before = Common::AddOutliersToCloud(before, 0);               // common.cpp:180-181 (no outliers: no draws)
    after = Common::AddOutliersToCloud(after, 0);
    const auto rotation = Tests::GetRandomRotationMatrix(rot_range);          // common.cpp:198
    const auto translation = Tests::GetRandomTranslationVector(trans_range);  // common.cpp:199
    after = Common::GetTransformedCloud(after, rotation, translation);        // common.cpp:201-204
    from_cloud(before, before_xyz);
    from_cloud(after, after_xyz);
    *nb_out = (int)before.size();
    *na_out = (int)after.size();
    from_mat3(rotation, rot9_colmajor);
    trans3[0] = translation.x; trans3[1] = translation.y; trans3[2] = translation.z;
}

// Consecutive Common::GetRandomPermutationVector draws (common.cpp:554-560) of the given sizes from a freshly seeded generator,
// concatenated: the generator stream GetCloudsFromConfig consumes (a std::shuffle of k elements advances it exactly like the
// permutation of size k does, and moves element perm[i] to place i).
REF_API void ref_permutation_sequence(unsigned seed, const int* sizes, int count, int* out)
{
    Common::mtRandom = std::mt19937{ seed };
    for (int i = 0; i < count; i++) {
        auto p = Common::GetRandomPermutationVector(sizes[i]);
        std::memcpy(out, p.data(), sizeof(int) * sizes[i]);
        out += sizes[i];
    }
}

// Common::GetCorrespondingPoints  common.cpp:509-515 (sequential :399-439, parallel :441-507).
// Returns the number of kept pairs; idx_before/idx_after hold the kept pair indices.
REF_API int ref_corresponding_points(const float* before_xyz, int n, const float* after_xyz, int m,
                                     float max_distance_squared, int parallel, int* idx_before, int* idx_after)
{
    auto t = Common::GetCorrespondingPoints(to_cloud(before_xyz, n), to_cloud(after_xyz, m),
                                            max_distance_squared, parallel != 0);
    const auto& ib = std::get<2>(t);
    const auto& ia = std::get<3>(t);
    std::memcpy(idx_before, ib.data(), ib.size() * sizeof(int));
    std::memcpy(idx_after, ia.data(), ia.size() * sizeof(int));
    return static_cast<int>(ib.size());
}

// Common::LeastSquaresSVD  common.cpp:517-552 on two clouds already in corresponding order.
REF_API void ref_least_squares_svd(const float* before_xyz, const float* after_xyz, int n, float* rot9_colmajor, float* trans3)
{
    auto r = Common::LeastSquaresSVD(to_cloud(before_xyz, n), to_cloud(after_xyz, n));
    from_mat3(r.first, rot9_colmajor);
    trans3[0] = r.second.x; trans3[1] = r.second.y; trans3[2] = r.second.z;
}

// Common::GetTransformedCloud (R, t)  common.cpp:219-224
REF_API void ref_transform_cloud(const float* xyz, int n, const float* rot9_colmajor, const float* trans3, float* out_xyz)
{
    auto c = Common::GetTransformedCloud(to_cloud(xyz, n), to_mat3(rot9_colmajor), glm::vec3(trans3[0], trans3[1], trans3[2]));
    from_cloud(c, out_xyz);
}

// Common::GetMeanSquaredError (indexed)  common.cpp:259-268
REF_API float ref_mse_indexed(const float* before_xyz, int n, const float* after_xyz, int m,
                              const int* idx_before, const int* idx_after, int k)
{
    std::vector<int> ib(idx_before, idx_before + k), ia(idx_after, idx_after + k);
    return Common::GetMeanSquaredError(to_cloud(before_xyz, n), to_cloud(after_xyz, m), ib, ia);
}

// BasicICP::GetBasicICPTransformationMatrix  cpu-slam/basicicp.cpp:23-61
REF_API void ref_icp(const float* before_xyz, int n, const float* after_xyz, int m, float eps,
                     float max_distance_squared, int max_iterations, int parallel,
                     float* rot9_colmajor, float* trans3, int* iterations, float* error)
{
    auto r = BasicICP::GetBasicICPTransformationMatrix(to_cloud(before_xyz, n), to_cloud(after_xyz, m), iterations, error,
                                                       eps, max_distance_squared, max_iterations, parallel != 0);
    from_mat3(r.first, rot9_colmajor);
    trans3[0] = r.second.x; trans3[1] = r.second.y; trans3[2] = r.second.z;
}

// CoherentPointDrift::CalculateSigmaSquared  cpu-slam/coherentpointdrift.cpp:126-139
REF_API float ref_cpd_sigma_squared(const float* before_xyz, int m, const float* after_xyz, int n)
{
    return CoherentPointDrift::CalculateSigmaSquared(to_cloud(before_xyz, m), to_cloud(after_xyz, n));
}

// CoherentPointDrift::ComputePMatrix (exact Gaussian E-step, no truncation)  cpu-slam/coherentpointdrift.cpp:168-221
// p1[m], pt1[n], px[m*3] row-major (px.row(k) = (x,y,z)), *L = Probabilities::error.
REF_API void ref_cpd_estep(const float* transformed_xyz, int m, const float* after_xyz, int n, float constant,
                           float sigma_squared, float* p1, float* pt1, float* px, float* L)
{
    auto p = CoherentPointDrift::ComputePMatrix(to_cloud(transformed_xyz, m), to_cloud(after_xyz, n), constant,
                                                sigma_squared, false, -1.0f);
    for (int k = 0; k < m; k++) {
        p1[k] = p.p1(k);
        for (int d = 0; d < 3; d++) px[3 * k + d] = p.px(k, d);
    }
    for (int x = 0; x < n; x++) pt1[x] = p.pt1(x);
    *L = p.error;
}

// CoherentPointDrift::MStep  cpu-slam/coherentpointdrift.cpp:223-277
// scale and sigma_squared are in/out exactly as in the reference (scale is left untouched when const_scale).
REF_API void ref_cpd_mstep(const float* before_xyz, int m, const float* after_xyz, int n, const float* p1,
                           const float* pt1, const float* px, int const_scale, float* rot9_colmajor, float* trans3,
                           float* scale, float* sigma_squared)
{
    CoherentPointDrift::Probabilities p;
    p.p1 = Eigen::VectorXf(m);
    p.pt1 = Eigen::VectorXf(n);
    p.px = Eigen::MatrixXf(m, 3);
    for (int k = 0; k < m; k++) {
        p.p1(k) = p1[k];
        for (int d = 0; d < 3; d++) p.px(k, d) = px[3 * k + d];
    }
    for (int x = 0; x < n; x++) p.pt1(x) = pt1[x];
    p.error = 0.f;
    glm::mat3 R = to_mat3(rot9_colmajor);
    glm::vec3 t(trans3[0], trans3[1], trans3[2]);
    CoherentPointDrift::MStep(to_cloud(before_xyz, m), to_cloud(after_xyz, n), p, const_scale != 0, &R, &t, scale, sigma_squared);
    from_mat3(R, rot9_colmajor);
    trans3[0] = t.x; trans3[1] = t.y; trans3[2] = t.z;
}

// CoherentPointDrift::GetRigidCPDTransformationMatrix  cpu-slam/coherentpointdrift.cpp:69-124
// fgt: 0 = None, 1 = Full, 2 = Hybrid (enumerators.h:18-23). Returns scale*R in rot9 (coherentpointdrift.cpp:123).
REF_API void ref_cpd(const float* before_xyz, int m, const float* after_xyz, int n, float eps, float weight,
                     int const_scale, int max_iterations, float tolerance, int fgt, float ratio_of_far_field,
                     float order_of_truncation, float* rot9_colmajor, float* trans3, int* iterations, float* error)
{
    auto r = CoherentPointDrift::GetRigidCPDTransformationMatrix(
        to_cloud(before_xyz, m), to_cloud(after_xyz, n), iterations, error, eps, weight, const_scale != 0,
        max_iterations, tolerance, static_cast<Common::ApproximationType>(fgt), ratio_of_far_field, order_of_truncation);
    from_mat3(r.first, rot9_colmajor);
    trans3[0] = r.second.x; trans3[1] = r.second.y; trans3[2] = r.second.z;
}

// CoherentPointDrift::ComputePMatrix with the truncated kernel of the hybrid mode (doTruncate = true)
// cpu-slam/coherentpointdrift.cpp:166 (call site) / :168-221
REF_API void ref_cpd_estep_truncated(const float* transformed_xyz, int m, const float* after_xyz, int n, float constant,
                                     float sigma_squared, float truncate, float* p1, float* pt1, float* px, float* L)
{
    auto p = CoherentPointDrift::ComputePMatrix(to_cloud(transformed_xyz, m), to_cloud(after_xyz, n), constant,
                                                sigma_squared, true, truncate);
    for (int k = 0; k < m; k++) {
        p1[k] = p.p1(k);
        for (int d = 0; d < 3; d++) px[3 * k + d] = p.px(k, d);
    }
    for (int x = 0; x < n; x++) pt1[x] = p.pt1(x);
    *L = p.error;
}

// CoherentPointDrift::ComputePMatrixWithFGT  common/cpdutils.cpp:19-77
REF_API void ref_cpd_estep_fgt(const float* transformed_xyz, int m, const float* after_xyz, int n, float weight,
                               float sigma_squared, float sigma_squared_init, float ratio_of_far_field,
                               float order_of_truncation, float* p1, float* pt1, float* px, float* L)
{
    auto p = CoherentPointDrift::ComputePMatrixWithFGT(to_cloud(transformed_xyz, m), to_cloud(after_xyz, n), weight,
                                                       sigma_squared, sigma_squared_init, ratio_of_far_field,
                                                       order_of_truncation);
    for (int k = 0; k < m; k++) {
        p1[k] = p.p1(k);
        for (int d = 0; d < 3; d++) px[3 * k + d] = p.px(k, d);
    }
    for (int x = 0; x < n; x++) pt1[x] = p.pt1(x);
    *L = p.error;
}

// FastGaussTransform::ComputeFGTModel  common/fgt.cpp:63-86: xc_out[3*K], ak_out[pd*K] column-major (pd rows).
// Returns pd.
REF_API int ref_fgt_model(const float* cloud_xyz, int n, const float* weights, float sigma, int K, int p,
                          float* xc_out, float* ak_out)
{
    std::vector<float> w(weights, weights + n);
    auto model = FastGaussTransform::ComputeFGTModel(to_cloud(cloud_xyz, n), w, sigma, K, p);
    from_cloud(model.xc, xc_out);
    const int pd = static_cast<int>(model.Ak.rows());
    std::memcpy(ak_out, model.Ak.data(), sizeof(float) * pd * K);
    return pd;
}

// FastGaussTransform::ComputeFGTPredict  common/fgt.cpp:88-150 on a model handed back in ref_fgt_model's layout.
REF_API void ref_fgt_predict(const float* cloud_xyz, int n, const float* xc, const float* ak, int pd, float sigma,
                             float e_param, int K, int p, float* v_out)
{
    FastGaussTransform::FGT_Model model;
    model.xc = to_cloud(xc, K);
    model.Ak = Eigen::Map<const Eigen::MatrixXf>(ak, pd, K);
    auto v = FastGaussTransform::ComputeFGTPredict(to_cloud(cloud_xyz, n), model, sigma, e_param, K, p);
    std::memcpy(v_out, v.data(), sizeof(float) * n);
}

// NonIterative::GetSingleNonIterativeSlamResult  cpu-slam/noniterative.cpp:25-55 (one PCA alignment of two equally ordered clouds)
REF_API void ref_nicp_single(const float* before_xyz, int m, const float* after_xyz, int n, float* rot9_colmajor, float* trans3,
                             float* approximated_error)
{
    auto r = NonIterative::GetSingleNonIterativeSlamResult(to_cloud(before_xyz, m), to_cloud(after_xyz, n));
    from_mat3(r.getRotationMatrix(), rot9_colmajor);
    const glm::vec3 t = r.getTranslationVector();
    trans3[0] = t.x; trans3[1] = t.y; trans3[2] = t.z;
    *approximated_error = r.getApproximatedError();
}

// NonIterative::GetNonIterativeTransformationMatrix  cpu-slam/noniterative.cpp:284-290, after seeding Common::mtRandom the way
// clouds-from-config does (common.cpp:136-137).  approximation: 0 none, 1 full, 2 hybrid.  parallel = 0: the sequential policy
// (:204-282), the only deterministic one (the parallel policy's threads race on the shared generator).
REF_API void ref_nicp(const float* before_xyz, int m, const float* after_xyz, int n, float eps, int max_repetitions,
                      int approximation, int parallel, int subcloud_size, unsigned seed, float* rot9_colmajor, float* trans3,
                      int* repetitions, float* error)
{
    Common::mtRandom = std::mt19937{ seed };
    auto r = NonIterative::GetNonIterativeTransformationMatrix(to_cloud(before_xyz, m), to_cloud(after_xyz, n), repetitions, error,
                                                               eps, max_repetitions, static_cast<Common::ApproximationType>(approximation),
                                                               parallel != 0, subcloud_size);
    from_mat3(r.first, rot9_colmajor);
    trans3[0] = r.second.x; trans3[1] = r.second.y; trans3[2] = r.second.z;
}

// The same call as a program of the reference makes it: BEHIND ref_clouds_from_config_* in the same process, on the generator as the input stage
// left it (mainwrapper.cpp: GetCloudsFromConfig, then the SlamFunc) -- no reseeding.
REF_API void ref_nicp_continue(const float* before_xyz, int m, const float* after_xyz, int n, float eps, int max_repetitions,
                               int approximation, int parallel, int subcloud_size, float* rot9_colmajor, float* trans3,
                               int* repetitions, float* error)
{
    auto r = NonIterative::GetNonIterativeTransformationMatrix(to_cloud(before_xyz, m), to_cloud(after_xyz, n), repetitions, error,
                                                               eps, max_repetitions, static_cast<Common::ApproximationType>(approximation),
                                                               parallel != 0, subcloud_size);
    from_mat3(r.first, rot9_colmajor);
    trans3[0] = r.second.x; trans3[1] = r.second.y; trans3[2] = r.second.z;
}

// Common::GetRandomPermutationVector  common.cpp:554-560 from a freshly seeded generator; `skip` permutations of the same size
// are drawn and dropped first (the driver draws one for the subcloud before the repetitions start).
REF_API void ref_random_permutation(unsigned seed, int size, int skip, int* out)
{
    Common::mtRandom = std::mt19937{ seed };
    for (int i = 0; i < skip; i++) (void)Common::GetRandomPermutationVector(size);
    auto p = Common::GetRandomPermutationVector(size);
    std::memcpy(out, p.data(), sizeof(int) * size);
}
