// The run configuration of the registration program: the same JSON keys, defaults and validation as the reference's
// parser (source/common/configparser.cpp:70-266, struct source/common/configuration.h:8-45), read by a small
// self-contained JSON reader (the reference uses the vendored nlohmann header).
//
// Quirks kept on purpose:
//   * the random-transform keys are "translation-range" / "rotation-range" (the schema file says "angle-range", the
//     parser reads "rotation-range": configparser.cpp:170);
//   * "cpd-const-scale" defaults to FALSE when parsed (configparser.cpp:240) although the struct default is true;
//   * "rotation" is 9 numbers, row-major in the file (configparser.cpp:139-141), multiplied by "scale" (:147);
//   * one of {translation + rotation} or {translation-range + rotation-range} is mandatory (:259-266).
#pragma once
#include <optional>
#include <string>
#include <utility>

#include "slam_types.h"

namespace Common {

enum class ComputationMethod { Icp, NoniterativeIcp, Cpd };
enum class ExecutionPolicy { Sequential, Parallel };
enum class ApproximationType { None, Full, Hybrid };

struct Configuration {
    // required
    ComputationMethod ComputationMethod_ = ComputationMethod::Icp;
    std::string BeforePath;
    std::string AfterPath;

    // optional
    std::optional<ExecutionPolicy> ExecutionPolicy_;
    std::optional<std::pair<Mat3, Vec3>> Transformation;               // rotation (already times scale), translation
    std::optional<std::pair<float, float>> TransformationParameters;   // rotation range, translation range
    std::optional<int> MaxIterations;
    std::optional<int> CloudBeforeResize;
    std::optional<int> CloudAfterResize;
    std::optional<float> CloudSpread;
    std::optional<int> RandomSeed;
    std::optional<float> NoiseAffectedPointsBefore;
    std::optional<float> NoiseAffectedPointsAfter;

    // optional with defaults (values as the PARSER leaves them)
    bool ShowVisualisation = false;
    float MaxDistanceSquared = 1000.f;
    ApproximationType ApproximationType_ = ApproximationType::Hybrid;
    int NicpBatchSize = 16;
    int NicpIterations = 32;
    int NicpSubcloudSize = 1000;
    float CpdWeight = .3f;
    bool CpdConstScale = false;
    float CpdTolerance = 1e-3f;
    float ConvergenceEpsilon = 1e-3f;
    float NoiseIntensityBefore = 0.1f;
    float NoiseIntensityAfter = 0.1f;
    int AdditionalOutliersBefore = 0;
    int AdditionalOutliersAfter = 0;
    float RatioOfFarField = 10.0f;
    int OrderOfTruncation = 8;

    void Print() const;
};

// Parses the JSON text; on any error prints "Parsing error: ..." like the reference and returns false.
bool ParseConfiguration(const std::string& json_text, Configuration* out);

// argv handling of ConfigParser::ConfigParser (configparser.cpp:11-39): no argument -> config/default.json, one argument ->
// that file if it exists else the default, more -> usage + default.
bool LoadConfigurationFromArgs(int argc, char** argv, Configuration* out);

}  // namespace Common
