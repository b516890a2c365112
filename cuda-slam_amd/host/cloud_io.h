// Input side of the registration program: OBJ point reader and the clouds-from-configuration stage that feeds the
// registration call (the reference's LoadCloud / GetCloudsFromConfig, source/common/common.cpp:16-23, :134-210).
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi_slam.h"
#include "configuration.h"
#include "slam_types.h"

namespace Common {

// One point per FACE CORNER in face order, polygons fan-triangulated; a file without faces yields its vertices.
// This is what the reference's assimp loader produces (aiProcess_Triangulate, no JoinIdenticalVertices:
// source/common/loader.cpp:30-67) -- bunny.obj (2 503 v, 4 968 f) -> 14 904 points (source/common/testset.cpp:22).
CpuCloud LoadCloud(const std::string& path);

CpuCloud NormalizeCloud(const CpuCloud& cloud, float size);                       // common.cpp:81-95
CpuCloud GetTransformedCloud(const CpuCloud& cloud, const Mat3& R, const Vec3& t);   // common.cpp:219-224

// Load, optional resize, normalise to "cloud-spread", shuffle with mt19937("random-seed"), optional noise / outliers,
// apply the configured transform to `after` (common.cpp:134-210).  Returns (before, after).
std::pair<CpuCloud, CpuCloud> GetCloudsFromConfig(const Configuration& config);

// The same stage, same results bit for bit, with the per-point work on the device (mi_prepare_cloud, include/mi_slam.h): the
// host only reads the files and draws the random outcomes from the program's generators, in the reference's order.
std::pair<CpuCloud, CpuCloud> GetCloudsFromConfigOnDevice(const Configuration& config, mi_ctx* ctx);

// iota + std::shuffle on the program's generator (the one "random-seed" seeds and the cloud stage has already drawn from), as
// Common::GetRandomPermutationVector does on Common::mtRandom (common.cpp:554-560)
std::vector<int> GetRandomPermutationVector(int size);

}  // namespace Common
