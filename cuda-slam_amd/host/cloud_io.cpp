#include "cloud_io.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <numeric>
#include <random>
#include <sstream>

namespace Common {
namespace {

std::mt19937 g_rng{0};

Point_f add(const Point_f& a, const Point_f& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Point_f sub(const Point_f& a, const Point_f& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Point_f mul(const Point_f& a, float s) { return {a.x * s, a.y * s, a.z * s}; }

// sequential fp32 accumulate, then divide (common.cpp:281-284)
Point_f center_of_mass(const CpuCloud& c)
{
    Point_f s;
    for (const auto& p : c) s = add(s, p);
    const float n = (float)c.size();
    return {s.x / n, s.y / n, s.z / n};
}

float spread_of(const CpuCloud& c)   // largest axis-aligned extent (common.cpp:57-79)
{
    if (c.empty()) return 0.f;
    Point_f lo = c[0], hi = c[0];
    for (const auto& p : c) {
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
    }
    return std::max({hi.x - lo.x, hi.y - lo.y, hi.z - lo.z});
}

std::pair<Point_f, Point_f> bounds_of(const CpuCloud& c)
{
    Point_f lo = c.empty() ? Point_f{} : c[0], hi = lo;
    for (const auto& p : c) {
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
    }
    return {lo, hi};
}

std::vector<int> random_permutation(int n)
{
    std::vector<int> p((size_t)n);
    std::iota(p.begin(), p.end(), 0);
    std::shuffle(p.begin(), p.end(), g_rng);
    return p;
}

float rand_float(float lo, float hi) { return static_cast<float>(rand()) / RAND_MAX * (hi - lo) + lo; }   // testutils.cpp:7-11
Point_f rand_point(const Point_f& lo, const Point_f& hi) { return {rand_float(lo.x, hi.x), rand_float(lo.y, hi.y), rand_float(lo.z, hi.z)}; }

CpuCloud subcloud(const CpuCloud& c, int size)   // common.cpp:25-37
{
    // `int >= size_t` in the reference: the int is converted, so a NEGATIVE size compares as huge and returns the cloud whole
    if ((size_t)size >= c.size()) return c;
    auto perm = random_permutation((int)c.size());
    perm.resize((size_t)size);
    CpuCloud out;
    out.reserve((size_t)size);
    for (int i : perm) out.push_back(c[(size_t)i]);
    return out;
}

CpuCloud add_noise(const CpuCloud& c, float share, float intensity)   // common.cpp:97-119
{
    CpuCloud out = c;
    const int n = (int)c.size();
    const int affected = std::clamp((int)std::round(share * n), 0, n);
    std::vector<char> flag((size_t)n, 0);
    std::fill(flag.begin(), flag.begin() + affected, 1);
    const auto perm = random_permutation(n);
    std::vector<char> mask((size_t)n);
    for (int i = 0; i < n; i++) mask[(size_t)i] = flag[(size_t)perm[(size_t)i]];
    const float reach = spread_of(c) * intensity;
    for (int i = 0; i < n; i++)
        if (mask[(size_t)i]) out[(size_t)i] = add(out[(size_t)i], rand_point({-reach, -reach, -reach}, {reach, reach, reach}));
    return out;
}

CpuCloud add_outliers(const CpuCloud& c, int count)   // common.cpp:121-132
{
    CpuCloud out = c;
    const auto [lo, hi] = bounds_of(c);
    for (int i = 0; i < count; i++) out.push_back(rand_point(lo, hi));
    return out;
}

Mat3 rotation_about(const Vec3& axis_in, float angle)   // glm::rotate(mat4(1), angle, normalize(axis)) as a 3x3
{
    // glm::normalize = v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt: a MULTIPLICATION by the reciprocal (func_geometric.inl:82-89,
    // func_exponential.inl:134-139) -- dividing by the length instead differs in the last bit of the axis and of every point of `after`
    // (round 5: found by the convergence set's fixtures, tests/test_noise_corpus.py)
    // ... and TWICE: the reference calls glm::rotate(mat4(1), angle, glm::normalize(axis)) (testutils.cpp:45-46) and glm::rotate normalises its axis
    // argument again (ext/matrix_transform.inl) -- the second pass moves last bits of the axis and 2-4 ulp of the off-diagonal entries
    const auto normalize = [](const Vec3& v) {
        const float inv = 1.f / std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
        return Vec3{v.x * inv, v.y * inv, v.z * inv};
    };
    const Vec3 a = normalize(normalize(axis_in));
    const float c = std::cos(angle), s = std::sin(angle), k = 1.f - c;
    Mat3 R;
    R[0][0] = c + k * a.x * a.x;       R[0][1] = k * a.x * a.y + s * a.z; R[0][2] = k * a.x * a.z - s * a.y;
    R[1][0] = k * a.y * a.x - s * a.z; R[1][1] = c + k * a.y * a.y;       R[1][2] = k * a.y * a.z + s * a.x;
    R[2][0] = k * a.z * a.x + s * a.y; R[2][1] = k * a.z * a.y - s * a.x; R[2][2] = c + k * a.z * a.z;
    return R;
}

}  // namespace

CpuCloud LoadCloud(const std::string& path)
{
    std::ifstream f(path);
    CpuCloud verts, corners;
    std::string line;
    while (std::getline(f, line)) {
        if (line.size() > 2 && line[0] == 'v' && line[1] == ' ') {
            std::istringstream ss(line.substr(2));
            Point_f p;
            ss >> p.x >> p.y >> p.z;
            verts.push_back(p);
        } else if (line.size() > 2 && line[0] == 'f' && line[1] == ' ') {
            std::istringstream ss(line.substr(2));
            std::vector<long> ids;
            std::string tok;
            while (ss >> tok) {
                const long i = std::strtol(tok.c_str(), nullptr, 10);   // "v", "v/vt", "v//vn", "v/vt/vn"
                ids.push_back(i > 0 ? i - 1 : (long)verts.size() + i);
            }
            // one point per corner of the face AS WRITTEN: assimp's OBJ importer makes a vertex per face-vertex reference, its triangulation
            // step only re-indexes them (a quad stays 4 points), and the reference's loader copies mesh->mVertices (loader.cpp:58-66):
            // bird.obj, 8 752 quads, is the 35 008 points of testset.cpp:25-26
            for (long id : ids)
                if (id >= 0 && id < (long)verts.size()) corners.push_back(verts[(size_t)id]);
        }
    }
    return corners.empty() ? verts : corners;
}

CpuCloud NormalizeCloud(const CpuCloud& cloud, float size)
{
    const Point_f center = center_of_mass(cloud);
    CpuCloud aligned(cloud.size());
    std::transform(cloud.begin(), cloud.end(), aligned.begin(), [&](const Point_f& p) { return sub(p, center); });
    const float extent = spread_of(aligned);
    if (std::abs(extent) < 1e-15) return cloud;
    const float scale = size / extent;
    const Point_f back = mul(center, -1.f);
    for (auto& p : aligned) p = sub(mul(p, scale), back);
    return aligned;
}

CpuCloud GetTransformedCloud(const CpuCloud& cloud, const Mat3& R, const Vec3& t)
{
    CpuCloud out(cloud.size());
    for (size_t i = 0; i < cloud.size(); i++) {
        const Vec3 r = R * Vec3{cloud[i].x, cloud[i].y, cloud[i].z};
        out[i] = {r.x + t.x, r.y + t.y, r.z + t.z};
    }
    return out;
}

std::pair<CpuCloud, CpuCloud> GetCloudsFromConfig(const Configuration& config)
{
    const unsigned seed = config.RandomSeed ? (unsigned)*config.RandomSeed : std::random_device{}();
    g_rng = std::mt19937{seed};

    const bool same = config.BeforePath == config.AfterPath;
    CpuCloud before = LoadCloud(config.BeforePath);
    CpuCloud after = same ? before : LoadCloud(config.AfterPath);

    if (config.CloudBeforeResize) before = subcloud(before, *config.CloudBeforeResize);
    if (config.CloudAfterResize) after = subcloud(after, *config.CloudAfterResize);
    if (config.CloudSpread) {
        before = NormalizeCloud(before, *config.CloudSpread);
        after = NormalizeCloud(after, *config.CloudSpread);
    }
    std::shuffle(before.begin(), before.end(), g_rng);
    std::shuffle(after.begin(), after.end(), g_rng);
    if (config.NoiseAffectedPointsBefore) before = add_noise(before, *config.NoiseAffectedPointsBefore, config.NoiseIntensityBefore);
    if (config.NoiseAffectedPointsAfter) after = add_noise(after, *config.NoiseAffectedPointsAfter, config.NoiseIntensityAfter);
    before = add_outliers(before, config.AdditionalOutliersBefore);
    after = add_outliers(after, config.AdditionalOutliersAfter);

    if (config.Transformation) {
        return {before, GetTransformedCloud(after, config.Transformation->first, config.Transformation->second)};
    }
    if (config.TransformationParameters) {
        // Tests::GetRandomRotationMatrix / GetRandomTranslationVector (source/common/testutils.cpp:43-55)
        const auto [rot_range, trans_range] = *config.TransformationParameters;
        const Point_f ax = rand_point({0, 0, 0}, {1, 1, 1});
        const Mat3 R = rotation_about({ax.x, ax.y, ax.z}, rot_range);
        const Point_f d = rand_point({-1, -1, -1}, {1, 1, 1});
        const float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        const Vec3 t{d.x / len * trans_range, d.y / len * trans_range, d.z / len * trans_range};
        return {before, GetTransformedCloud(after, R, t)};
    }
    return {before, after};
}

// The same stage with the per-point work on the device (mi_prepare_cloud): this function only draws the random OUTCOMES, each
// generator consumed in the order GetCloudsFromConfig consumes it -- g_rng: subcloud before / after, shuffle before / after,
// noise flags before / after; rand(): noise draws before / after, outlier draws before / after, then the random transform.
std::pair<CpuCloud, CpuCloud> GetCloudsFromConfigOnDevice(const Configuration& config, mi_ctx* ctx)
{
    const unsigned seed = config.RandomSeed ? (unsigned)*config.RandomSeed : std::random_device{}();
    g_rng = std::mt19937{seed};

    const bool same = config.BeforePath == config.AfterPath;
    const CpuCloud raw_before = LoadCloud(config.BeforePath);
    const CpuCloud raw_after_file = same ? CpuCloud{} : LoadCloud(config.AfterPath);
    const CpuCloud& raw_after = same ? raw_before : raw_after_file;
    if (raw_before.empty() || raw_after.empty()) return {};

    struct Side {
        const CpuCloud* raw;
        int n;                                     // size after the optional resize
        std::vector<int> sub, shuffle, rows;
        std::vector<float> noise_unit, outlier_unit;
        float intensity = 0.f;
    } b{&raw_before, (int)raw_before.size(), {}, {}, {}, {}, {}}, a{&raw_after, (int)raw_after.size(), {}, {}, {}, {}, {}};

    const auto draw_subcloud = [](Side& s, const std::optional<int>& resize) {
        // GetSubcloud returns the cloud and draws nothing (its comparison is int against size_t: a negative value counts as huge)
        if (!resize || (size_t)*resize >= s.raw->size()) return;
        s.sub = random_permutation((int)s.raw->size());
        s.sub.resize((size_t)std::max(*resize, 0));
        s.n = (int)s.sub.size();
    };
    draw_subcloud(b, config.CloudBeforeResize);
    draw_subcloud(a, config.CloudAfterResize);
    if (b.n < 1 || a.n < 1) return {};
    b.shuffle = random_permutation(b.n);                               // std::shuffle of n elements == this permutation as a gather
    a.shuffle = random_permutation(a.n);
    const auto draw_flags = [](Side& s, const std::optional<float>& share, float intensity) {
        if (!share) return;
        const int affected = std::clamp((int)std::round(*share * s.n), 0, s.n);
        const auto perm = random_permutation(s.n);
        for (int i = 0; i < s.n; i++)
            if (perm[(size_t)i] < affected) s.rows.push_back(i);       // ApplyPermutation: flag[perm[i]]
        s.intensity = intensity;
    };
    draw_flags(b, config.NoiseAffectedPointsBefore, config.NoiseIntensityBefore);
    draw_flags(a, config.NoiseAffectedPointsAfter, config.NoiseIntensityAfter);
    const auto unit = [](size_t count) {                               // static_cast<float>(rand()) / RAND_MAX, testutils.cpp:10
        std::vector<float> u(count);
        for (auto& v : u) v = static_cast<float>(rand()) / RAND_MAX;
        return u;
    };
    b.noise_unit = unit(3 * b.rows.size());
    a.noise_unit = unit(3 * a.rows.size());
    b.outlier_unit = unit(3 * (size_t)std::max(config.AdditionalOutliersBefore, 0));
    a.outlier_unit = unit(3 * (size_t)std::max(config.AdditionalOutliersAfter, 0));

    mi_prepare_params pb;
    mi_prepare_params_default(&pb);
    pb.has_spread = config.CloudSpread ? 1 : 0;
    pb.spread = config.CloudSpread ? *config.CloudSpread : 1.f;
    mi_prepare_params pa = pb;
    pb.noise_intensity = b.intensity;
    pa.noise_intensity = a.intensity;
    Mat3 R;
    Vec3 t;
    if (config.Transformation) {
        R = config.Transformation->first;
        t = config.Transformation->second;
        pa.has_transform = 1;
    } else if (config.TransformationParameters) {
        const auto [rot_range, trans_range] = *config.TransformationParameters;
        const Point_f ax = rand_point({0, 0, 0}, {1, 1, 1});
        R = rotation_about({ax.x, ax.y, ax.z}, rot_range);
        const Point_f d = rand_point({-1, -1, -1}, {1, 1, 1});
        const float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        t = Vec3{d.x / len * trans_range, d.y / len * trans_range, d.z / len * trans_range};
        pa.has_transform = 1;
    }
    for (int k = 0; k < 9; k++) pa.rotation[k] = R.data()[k];
    for (int k = 0; k < 3; k++) pa.translation[k] = t[k];

    const auto run = [ctx](const Side& s, const mi_prepare_params& p, CpuCloud* out) {
        const int n_out = (int)s.outlier_unit.size() / 3;
        out->resize((size_t)s.n + (size_t)n_out);
        int got = 0;
        const int rc = mi_prepare_cloud(ctx, &(*s.raw)[0].x, (int)s.raw->size(), s.sub.empty() ? nullptr : s.sub.data(), s.n, s.shuffle.data(),
                                        s.rows.empty() ? nullptr : s.rows.data(), s.noise_unit.empty() ? nullptr : s.noise_unit.data(),
                                        (int)s.rows.size(), n_out ? s.outlier_unit.data() : nullptr, n_out, &p, &(*out)[0].x, &got);
        if (rc != MI_OK) {       // checkCudaErrors behaviour (include/helper_cuda.h:567-573): report and leave
            fprintf(stderr, "MI355X device error at mi_prepare_cloud: code=%d \"%s\"\n", rc, mi_last_error());
            exit(EXIT_FAILURE);
        }
        out->resize((size_t)got);
    };
    CpuCloud before, after;
    run(b, pb, &before);
    run(a, pa, &after);
    return {before, after};
}

std::vector<int> GetRandomPermutationVector(int size) { return random_permutation(size); }

}  // namespace Common
