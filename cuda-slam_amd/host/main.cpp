// mi-slam: the reference's `cuda-slam` program (source/cuda-slam/gpumain.cpp:50-56 + Common::Main,
// source/common/mainwrapper.cpp:5-54) on the MI355X path: parse the JSON configuration, build the two clouds, call the
// SlamFunc, print rotation / translation / error in the reference's format.  The OpenGL viewer is not part of this build.
//
//   mi-slam [config.json] [--rules cuda|cpu] [--device N] [--prepare device|host] [--dump-clouds file.bin] [--result-json file.json]
//
// --prepare: where the clouds-from-configuration stage runs its per-point work: on the device (mi_prepare_cloud, the default) or
// on the host (cloud_io.cpp's mirror of the reference) -- identical clouds either way.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "cloud_io.h"
#include "configuration.h"
#include "slam_adapter.h"

using namespace Common;

static void print_matrix3(const Mat3& m)   // PrintMatrixWithSize(matrix, 3), common.cpp:374-381
{
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 3; i++) printf("%1.8f ", m[i][j]);
        printf("\n");
    }
}

int main(int argc, char** argv)
{
    std::vector<char*> positional{argv[0]};
    std::string dump_path, json_path, rules = "cuda", prepare = "device";
    int device = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--rules") && i + 1 < argc) rules = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--prepare") && i + 1 < argc) prepare = argv[++i];
        else if (!strcmp(argv[i], "--dump-clouds") && i + 1 < argc) dump_path = argv[++i];
        else if (!strcmp(argv[i], "--result-json") && i + 1 < argc) json_path = argv[++i];
        else positional.push_back(argv[i]);
    }

    Configuration configuration;
    if (!LoadConfigurationFromArgs((int)positional.size(), positional.data(), &configuration)) {
        printf("Aborting\n");
        return -1;
    }
    configuration.Print();
    const unsigned seed = configuration.RandomSeed ? (unsigned)*configuration.RandomSeed : (unsigned)time(nullptr);
    SetSlamDevice(device);
    // the context first: bringing up the HIP runtime consumes rand() draws of its own, and the stage below must see the
    // stream exactly as srand(seed) leaves it
    mi_ctx* prepare_ctx = prepare == "host" ? nullptr : GetSlamContext();
    srand(seed);                                                        // mainwrapper.cpp:17-18

    auto [before, after] = prepare_ctx ? GetCloudsFromConfigOnDevice(configuration, prepare_ctx) : GetCloudsFromConfig(configuration);
    if (before.empty() || after.empty()) {
        printf("Aborting: empty cloud (before %zu, after %zu points)\n", before.size(), after.size());
        return -1;
    }
    if (!dump_path.empty()) {   // int32 n, int32 m, n*3 floats, m*3 floats -- lets a test compare the input stage
        FILE* f = fopen(dump_path.c_str(), "wb");
        if (!f) { perror("dump-clouds"); return -1; }
        const int n = (int)before.size(), m = (int)after.size();
        fwrite(&n, 4, 1, f); fwrite(&m, 4, 1, f);
        fwrite(before.data(), sizeof(Point_f), before.size(), f);
        fwrite(after.data(), sizeof(Point_f), after.size(), f);
        fclose(f);
        if (json_path.empty() && getenv("MISLAM_DUMP_ONLY")) return 0;
    }

    SetSlamRules(rules == "cpu" ? SlamRules::CpuSlam : SlamRules::CudaSlam, configuration.MaxDistanceSquared);
    const SlamFunc func = GetGpuSlamResult;
    int iterations = 0;
    float error = 0.f;
    const auto result = func(before, after, configuration, &iterations, &error);   // mainwrapper.cpp:25

    const Vec3 vec = result.second;
    printf("Results:\n");
    printf("Rotation matrix:\n");
    print_matrix3(result.first);
    printf("Translation vector:\n");
    printf("x = %f, y = %f, z = %f\n", vec.x, vec.y, vec.z);
    printf("Error: %f\n", error);

    if (!json_path.empty()) {
        FILE* f = fopen(json_path.c_str(), "w");
        if (!f) { perror("result-json"); return -1; }
        fprintf(f, "{\"iterations\": %d, \"error\": %.9g, \"t\": [%.9g, %.9g, %.9g], \"R_colmajor\": [", iterations, error, vec.x, vec.y, vec.z);
        for (int i = 0; i < 9; i++) fprintf(f, "%s%.9g", i ? ", " : "", result.first.data()[i]);
        fprintf(f, "]}\n");
        fclose(f);
    }
    return 0;
}
